// C-ABI entry points of the fused AWP sample-feature embedding: reference networks/dpnerf/awp.py:36-37 (the module list),
// :98-100 (its forward), trained by run_nerf.py:593-601.
#include "awp_embed.h"
#include "evd_common.h"
#include "mlp_pipe.h"
#include "nerf_mlp.h"
#include "pack.h"
#include "voxel_train.h"

#include <cstdint>

using namespace evd;

struct evd_awp_embed {
    long param_off[2 * AWP_D + 1];           // W0, b0, W1, b1, ... in the flat parameter arena; [8] = total
    PackedStream fwd[2], bwd[2][AWP_D];      // [0] bf16, [1] f16
    DevBuf bias, bias_src, maps;
    RepackBatch batch;
};

static int prec_index(int precision) { return precision == EVD_PREC_BF16 ? 0 : (precision == EVD_PREC_F16 ? 1 : -1); }

extern "C" {

void evd_awp_embed_destroy(evd_awp_embed* a) {
    if (!a) return;
    for (int i = 0; i < 2; ++i) {
        a->fwd[i].release();
        for (int l = 0; l < AWP_D; ++l) a->bwd[i][l].release();
    }
    a->bias.release(); a->bias_src.release(); a->maps.release();
    a->batch.release();
    delete a;
}

int evd_awp_embed_create(const float* const* weights, const float* const* biases, int input_ch, int width, int depth, evd_awp_embed** out) {
    EVD_REQUIRE(weights && biases && out, "evd_awp_embed_create: null argument");
    EVD_REQUIRE(input_ch == AWP_IN && width == AWP_W && depth == AWP_D,
                "evd_awp_embed_create: built for input_ch %d, W_sam %d, D_sam %d (fine_geo_feat_dim / kernel_awp_sam_emb_* of the shipped configs), got %d / %d / %d",
                AWP_IN, AWP_W, AWP_D, input_ch, width, depth);
    for (int l = 0; l < AWP_D; ++l) EVD_REQUIRE(weights[l] && biases[l], "evd_awp_embed_create: missing layer %d", l);
    evd_awp_embed* a = new evd_awp_embed();
    long total = 0;
    for (int l = 0; l < AWP_D; ++l) {
        a->param_off[2 * l] = total; total += (long)AWP_W * (l == 0 ? AWP_IN : AWP_W);
        a->param_off[2 * l + 1] = total; total += AWP_W;
    }
    a->param_off[2 * AWP_D] = total;
    std::vector<float> arena((size_t)total);
    for (int l = 0; l < AWP_D; ++l) {
        memcpy(arena.data() + a->param_off[2 * l], weights[l], sizeof(float) * AWP_W * (l == 0 ? AWP_IN : AWP_W));
        memcpy(arena.data() + a->param_off[2 * l + 1], biases[l], sizeof(float) * AWP_W);
    }
    const float* A = arena.data();
    auto hid_col = [](int j, int kk) { return 16 * j + phi(kk); };
    int rc = EVD_OK;
    for (int i = 0; i < 2 && !rc; ++i) {
        const int prec = i == 0 ? EVD_PREC_BF16 : EVD_PREC_F16;
        {   // forward stream, the layer table of awp_embed_kernel.h AwpNet: single-tile groups, 16 KiB chunks
            StreamBuilder sb(prec, PIPE_CB);
            sb.arena = A;
            sb.group = 1;
            for (int l = 0; l < AWP_D; ++l) {
                const int in_dim = l == 0 ? AWP_IN : AWP_W;
                sb.layer(A + a->param_off[2 * l], AWP_W, in_dim, AWP_W / 32, in_dim / 16, l == AWP_D - 1, hid_col);
            }
            if ((long)sb.bytes.size() != (long)AWP_NCHUNKS * PIPE_CB) { evd_awp_embed_destroy(a); return fail(EVD_E_INVALID, "evd_awp_embed_create: stream geometry"); }
            rc = a->fwd[i].upload(sb);
        }
        for (int l = 0; l < AWP_D && !rc; ++l) {        // W_l^T streams of the dgrad chain
            const int in_dim = l == 0 ? AWP_IN : AWP_W;
            StreamBuilder sb(prec, PIPE_CB);
            sb.arena = A;
            sb.group = 1;
            sb.layer_transposed(A + a->param_off[2 * l], AWP_W, in_dim, 0, in_dim, nullptr, 0, in_dim / 32, AWP_W / 16, true, hid_col);
            rc = a->bwd[i][l].upload(sb);
        }
    }
    if (!rc) {
        std::vector<float> b((size_t)AWP_D * AWP_W);
        std::vector<int32_t> bsrc(b.size());
        for (int l = 0; l < AWP_D; ++l)
            for (int c = 0; c < AWP_W; ++c) {
                b[(size_t)l * AWP_W + c] = A[a->param_off[2 * l + 1] + c];
                bsrc[(size_t)l * AWP_W + c] = (int32_t)(a->param_off[2 * l + 1] + c);
            }
        rc = a->bias.upload(b.data(), b.size() * sizeof(float));
        if (!rc) rc = a->bias_src.upload(bsrc.data(), bsrc.size() * sizeof(int32_t));
    }
    if (!rc) {
        std::vector<int> m(AMAP_TOTAL, -1);
        for (int i = 0; i < AWP_W; ++i) m[AMAP_H + i] = hid_col(i / 16, i % 16);
        for (int i = 0; i < AWP_IN; ++i) m[AMAP_GEO + i] = hid_col(i / 16, i % 16);
        rc = a->maps.upload(m.data(), m.size() * sizeof(int));
    }
    if (rc) { evd_awp_embed_destroy(a); return rc; }
    *out = a;
    return EVD_OK;
}

long evd_awp_embed_param_count(const evd_awp_embed* a) { return a ? a->param_off[2 * AWP_D] : 0; }

int evd_awp_embed_load_params(evd_awp_embed* a, const float* params, void* stream) {
    EVD_REQUIRE(a && params, "evd_awp_embed_load_params: null argument");
    hipStream_t st = as_stream(stream);
    int rc;
    std::vector<PackedStream*> all;
    for (int i = 0; i < 2; ++i) {
        all.push_back(&a->fwd[i]);
        for (int l = 0; l < AWP_D; ++l) all.push_back(&a->bwd[i][l]);
    }
    if ((rc = repack_batch(a->batch, all, params, st))) return rc;
    const long nb = (long)(a->bias.bytes / sizeof(float));
    hipLaunchKernelGGL(k_gather_f32, dim3((unsigned)cdiv(nb, 256L)), dim3(256), 0, st, params, (const int*)a->bias_src.p, nb, (float*)a->bias.p);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

size_t evd_awp_embed_store_bytes(const evd_awp_embed* a, long nsamp) {
    return (!a || nsamp < 0) ? 0 : (size_t)awp_tiles(nsamp) * awpstore::TILE_BYTES + awpstore::TRAILER_BYTES;
}

static const int AWP_WGRAD_BLOCKS = 256;
size_t evd_awp_embed_backward_workspace_bytes(void) { return (size_t)AWP_WGRAD_BLOCKS * 13 * 4096 + 512; }     // the fused backward's 13 blocks per workgroup (awp_bwd_fused.h); a per-layer wgrad needs 2 x 5

int evd_awp_embed_forward(const evd_awp_embed* a, int precision, const float* geo_rows, const evd_voxel* fine, const void* fine_store,
                          size_t fine_store_bytes, long nsamp, float* h_local, void* store, size_t store_bytes, void* stream) {
    EVD_REQUIRE(a && h_local && nsamp >= 0, "evd_awp_embed_forward: null argument");
    const int pi = prec_index(precision);
    EVD_REQUIRE(pi >= 0, "evd_awp_embed_forward: built for precision f16 / bf16");
    EVD_REQUIRE((geo_rows != nullptr) != (fine_store != nullptr), "evd_awp_embed_forward: pass the geo features EITHER as float32 rows OR as the fine level's store");
    if (nsamp == 0) return EVD_OK;
    AwpFwdParams p;
    p.wstream = (const char*)a->fwd[pi].data.p; p.bias = (const float*)a->bias.p; p.geo_rows = geo_rows;
    p.geo_frags = nullptr; p.geo_tile_bytes = 0; p.geo_slot = 0;
    if (fine_store) {
        EVD_REQUIRE(fine, "evd_awp_embed_forward: the fine level's store needs the level handle");
        const size_t need = evd_voxel_train_store_bytes(fine, nsamp);      // half-precision layout: the level must have run in THIS precision
        EVD_REQUIRE(evd_voxel_geo_feat_dim(fine) == AWP_IN, "evd_awp_embed_forward: the level has %d geo channels, the embedding reads %d", evd_voxel_geo_feat_dim(fine), AWP_IN);
        if (fine_store_bytes < need) return fail(EVD_E_WORKSPACE, "evd_awp_embed_forward: fine store %zu < %zu bytes", fine_store_bytes, need);
        p.geo_frags = (const char*)fine_store; p.geo_tile_bytes = voxel_store_tile_bytes(256); p.geo_slot = voxel_store_geo_slot(256);
    }
    p.nsamp = nsamp; p.h_local = h_local; p.act = (char*)store; p.nchunks = AWP_NCHUNKS;
    if (store && store_bytes < evd_awp_embed_store_bytes(a, nsamp))
        return fail(EVD_E_WORKSPACE, "evd_awp_embed_forward: store %zu < %zu bytes", store_bytes, evd_awp_embed_store_bytes(a, nsamp));
    return precision == EVD_PREC_F16 ? launch_awp_embed_f16(store != nullptr, p, as_stream(stream)) : launch_awp_embed_bf16(store != nullptr, p, as_stream(stream));
}

int evd_awp_embed_backward(const evd_awp_embed* a, int precision, const float* d_h_local, long nsamp, void* store, size_t store_bytes,
                           const evd_awp_embed_grads* grads, float* d_geo_rows, const unsigned* d_h_absmax, void* workspace,
                           size_t workspace_bytes, void* stream) {
    EVD_REQUIRE(a && d_h_local && store && grads && workspace && nsamp >= 0, "evd_awp_embed_backward: null argument");
    const int pi = prec_index(precision);
    EVD_REQUIRE(pi >= 0, "evd_awp_embed_backward: built for precision f16 / bf16");
    if (nsamp == 0) return EVD_OK;
    if (store_bytes < evd_awp_embed_store_bytes(a, nsamp)) return fail(EVD_E_WORKSPACE, "evd_awp_embed_backward: store %zu < %zu bytes", store_bytes, evd_awp_embed_store_bytes(a, nsamp));
    if (workspace_bytes < evd_awp_embed_backward_workspace_bytes())
        return fail(EVD_E_WORKSPACE, "evd_awp_embed_backward: workspace %zu < %zu bytes", workspace_bytes, evd_awp_embed_backward_workspace_bytes());
    AwpBwdPlan b;
    b.d_h_local = d_h_local; b.d_h_absmax = d_h_absmax; b.nsamp = nsamp; b.tiles = awp_tiles(nsamp); b.store = (char*)store;
    for (int l = 0; l < AWP_D; ++l) {
        b.wt[l] = (const char*)a->bwd[pi][l].data.p;
        b.grads.w[l] = grads->w[l]; b.grads.b[l] = grads->b[l];
    }
    b.maps = (const int*)a->maps.p;
    b.partial = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    b.wgrad_blocks = AWP_WGRAD_BLOCKS;
    b.d_geo_rows = d_geo_rows;
    return precision == EVD_PREC_F16 ? run_awp_backward_f16(b, as_stream(stream)) : run_awp_backward_bf16(b, as_stream(stream));
}

}  // extern "C"
