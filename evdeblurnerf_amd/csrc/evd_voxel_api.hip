// C-ABI entry points of the PDRF backbone and the mode='c2f' renderer (reference networks/pdrf/voxnerf.py,
// networks/renderer.py:182-217).
#include "evd_common.h"
#include "nerf_mlp.h"
#include "pack.h"
#include "voxel.h"
#include "voxel_mlp_kernel.h"
#include "voxel_train.h"
#include "awp_embed.h"

#include <cmath>
#include <cstdint>
#include <cstdlib>

// the PDRF level networks keep generic / pipelined / training streams for the first four arithmetic modes; EVD_PREC_F16C (compensated
// float16) is an INFERENCE mode with its own stream (pipe_c) on the fine level; the 64-wide coarse level runs EVD_PREC_F16X3 next to it
#define EVD_VOX_NUM_PREC EVD_PREC_F16C

using namespace evd;

struct evd_voxel {
    int num_layers, hidden_dim, geo, num_layers_color, input_ch, ft_dim, app_dim;
    int n_comp[3], grid[3], app_act, rgb_act, sigma_act, composite_feature;
    float aabb[6], rmnear;
    int multires = PE_L, multires_views = PE_LV;
    DevBuf plane[3], line[3], plane_h[3], line_h[3], basis, bias, bias_src, tv_acc, wmaps;
    PackedStream stream[EVD_VOX_NUM_PREC], pipe[EVD_VOX_NUM_PREC];    // pipe: stream of the software-pipelined kernel, where built
    int nchunks[EVD_VOX_NUM_PREC], pipe_chunks[EVD_VOX_NUM_PREC];
    // training path (bf16 / f16): the level's network on the software pipeline (the fine level shares `pipe`) and its W^T streams
    PackedStream train[EVD_VOX_NUM_PREC], bwd[EVD_VOX_NUM_PREC][VBWD_NSTREAMS];
    int train_chunks[EVD_VOX_NUM_PREC];
    mutable PackedStreamC pipe_c; // compensated float16 mode (voxel_mlp_c_kernel.h): float16 + fp6 fragments and row scales; fine level only
    int pipe_c_chunks = 0;
    // its re-pack after evd_voxel_load_params is LAZY (a training loop re-loads every iteration and never renders in this mode): the new
    // parameter values are kept in `arena_dev` (one stream-ordered device copy) and re-packed by the first f16c launch that follows
    mutable DevBuf arena_dev;
    mutable bool pipe_c_stale = false;
    DevBuf cnet;                  // composite_feature levels: color_net.{0,1,2}.{weight,bias} as float32 in the reference layouts (k_color_rows)
    long param_off[9];            // sigma_net.0, sigma_net.1, color_net.{0,1,2}.{weight,bias} in the parameter arena, [8] = total
    GridParams gp;
    RepackBatch batch;            // table of every fragment stream, for the one-launch re-pack of evd_voxel_load_params
    mutable SideStream side;      // backward entry: wgrad side stream of THIS handle (created on first use)
};

static const int kMat0[3] = {0, 0, 1}, kMat1[3] = {1, 2, 2}, kVec[3] = {2, 1, 0};

extern "C" {

void evd_voxel_destroy(evd_voxel* v) {
    if (!v) return;
    for (int i = 0; i < 3; ++i) { v->plane[i].release(); v->line[i].release(); v->plane_h[i].release(); v->line_h[i].release(); }
    for (int i = 0; i < EVD_VOX_NUM_PREC; ++i) {
        v->stream[i].release(); v->pipe[i].release(); v->train[i].release();
        for (int k = 0; k < VBWD_NSTREAMS; ++k) v->bwd[i][k].release();
    }
    v->basis.release(); v->bias.release(); v->bias_src.release(); v->tv_acc.release(); v->wmaps.release();
    v->pipe_c.release(); v->arena_dev.release(); v->cnet.release();
    v->side.release();
    v->batch.release();
    delete v;
}

int evd_voxel_create(const evd_voxel_desc* d, evd_voxel** out) {
    EVD_REQUIRE(d && out, "evd_voxel_create: null argument");
    // frequency counts other than (PE_L, PE_LV) run in the generic kernel only (inference; no pipelined / f16c / training streams)
    EVD_REQUIRE(d->multires >= 0 && d->multires <= PE_L_MAX && d->multires_views >= 0 && d->multires_views <= PE_LV,
                "evd_voxel_create: multires %d / multires_views %d out of range (0..%d / 0..%d)", d->multires, d->multires_views, PE_L_MAX, PE_LV);
    const int L = d->multires, Lv = d->multires_views;
    const bool standard = L == PE_L && Lv == PE_LV;
    EVD_REQUIRE(d->num_layers == 2 && d->num_layers_color == 3, "evd_voxel_create: only 2 sigma + 3 colour layers are built (all shipped configs)");
    const int IC = 3 * (1 + 2 * L), ICV = 3 * (1 + 2 * Lv);
    const int FT = d->input_ch - IC, HD = d->hidden_dim, G = d->geo_feat_dim;
    EVD_REQUIRE((HD == 64 && G == 15 && FT == 32) || (HD == 256 && G == 128 && FT == 64),
                "evd_voxel_create: (hidden %d, geo %d, features %d) not built: coarse 64/15/32 or fine 256/128/64", HD, G, FT);
    const int ctot = d->n_comp[0] + d->n_comp[1] + d->n_comp[2];
    EVD_REQUIRE(ctot <= 128 && d->n_comp[0] % 4 == 0 && d->n_comp[1] % 4 == 0 && d->n_comp[2] % 4 == 0 && d->app_dim <= 64,
                "evd_voxel_create: n_comp must be multiples of 4 with sum <= 128, app_dim <= 64");
    for (int i = 0; i < 3; ++i) EVD_REQUIRE(d->plane[i] && d->line[i] && d->grid[i] >= 1, "evd_voxel_create: missing grid %d", i);
    EVD_REQUIRE(d->basis && d->sigma_w[0] && d->sigma_w[1] && d->color_w[0] && d->color_w[1] && d->color_w[2], "evd_voxel_create: missing weights");

    evd_voxel* v = new evd_voxel();
    v->num_layers = 2; v->hidden_dim = HD; v->geo = G; v->num_layers_color = 3; v->input_ch = d->input_ch; v->ft_dim = FT;
    v->app_dim = d->app_dim; v->app_act = d->app_act; v->rgb_act = d->rgb_act; v->sigma_act = d->sigma_act;
    v->composite_feature = d->composite_feature ? 1 : 0; v->rmnear = d->rmnear;
    v->multires = L; v->multires_views = Lv;
    memcpy(v->aabb, d->aabb, sizeof(v->aabb));
    int rc = EVD_OK;
    // planes/lines: [1,C,H,W] -> channel-last [H][W][C]
    for (int i = 0; i < 3 && !rc; ++i) {
        v->n_comp[i] = d->n_comp[i]; v->grid[i] = d->grid[i];
        const int C = d->n_comp[i], Wp = d->grid[kMat0[i]], Hp = d->grid[kMat1[i]], Lp = d->grid[kVec[i]];
        std::vector<float> cl((size_t)C * Hp * Wp);
        for (int c = 0; c < C; ++c)
            for (long hw = 0; hw < (long)Hp * Wp; ++hw) cl[(size_t)hw * C + c] = d->plane[i][(size_t)c * Hp * Wp + hw];
        rc = v->plane[i].upload(cl.data(), cl.size() * sizeof(float));
        if (rc) break;
        {   // float16 copy for the half-precision arithmetic modes
            std::vector<_Float16> ch(cl.size());
            for (size_t k = 0; k < cl.size(); ++k) ch[k] = (_Float16)cl[k];
            rc = v->plane_h[i].upload(ch.data(), ch.size() * sizeof(_Float16));
            if (rc) break;
        }
        std::vector<float> ll((size_t)C * Lp);
        for (int c = 0; c < C; ++c)
            for (int l = 0; l < Lp; ++l) ll[(size_t)l * C + c] = d->line[i][(size_t)c * Lp + l];
        rc = v->line[i].upload(ll.data(), ll.size() * sizeof(float));
        if (rc) break;
        std::vector<_Float16> lh(ll.size());
        for (size_t k = 0; k < ll.size(); ++k) lh[k] = (_Float16)ll[k];
        rc = v->line_h[i].upload(lh.data(), lh.size() * sizeof(_Float16));
    }
    if (!rc) rc = v->basis.upload(d->basis, sizeof(float) * (size_t)d->app_dim * ctot);
    if (!rc) rc = v->tv_acc.alloc((size_t)6 * 2 * TV_MAX_BLOCKS * sizeof(double));
    if (rc) { evd_voxel_destroy(v); return rc; }
    GridParams& g = v->gp;
    for (int i = 0; i < 3; ++i) {
        g.plane[i] = (const float*)v->plane[i].p; g.line[i] = (const float*)v->line[i].p;
        g.plane_h[i] = (const _Float16*)v->plane_h[i].p; g.line_h[i] = (const _Float16*)v->line_h[i].p;
        g.n_comp[i] = d->n_comp[i]; g.grid[i] = d->grid[i];
        g.aabb_min[i] = d->aabb[i];
        g.inv[i] = 2.0f / (d->aabb[3 + i] - d->aabb[i]);
    }
    g.basis = (const float*)v->basis.p; g.app_dim = d->app_dim; g.app_act = d->app_act;

    // weight streams (order = kernel_voxel.hip k_voxel_mlp).  The parameters are first copied into one host arena in the
    // canonical order sigma_net.0, sigma_net.1, color_net.{0,1,2}.{weight,bias}; every packed element records its arena index so that
    // evd_voxel_load_params can re-pack on the device (pack.h).
    const int T = HD / 32, KS = HD / 16, KF = FT / 16;
    const long psz[8] = {(long)HD * d->input_ch, (long)(1 + G) * HD, (long)HD * (G + ICV), HD, (long)HD * HD, HD, 3L * HD, 3};
    long total = 0;
    for (int i = 0; i < 8; ++i) { v->param_off[i] = total; total += psz[i]; }
    v->param_off[8] = total;
    std::vector<float> arena((size_t)total, 0.f);
    {
        const float* srcs[8] = {d->sigma_w[0], d->sigma_w[1], d->color_w[0], d->color_b[0], d->color_w[1], d->color_b[1], d->color_w[2], d->color_b[2]};
        for (int i = 0; i < 8; ++i)
            if (srcs[i]) memcpy(arena.data() + v->param_off[i], srcs[i], psz[i] * sizeof(float));
    }
    if (v->composite_feature) {       // the colour network runs per RAY on the composited features (voxnerf.py:231-239): plain float32 rows
        rc = v->cnet.upload(arena.data() + v->param_off[2], (size_t)(v->param_off[8] - v->param_off[2]) * sizeof(float));
        if (rc) { evd_voxel_destroy(v); return rc; }
    }
    const float* A = arena.data();
    const float *sigma_w0 = A + v->param_off[0], *sigma_w1 = A + v->param_off[1], *color_w0 = A + v->param_off[2], *color_w1 = A + v->param_off[4],
                *color_w2 = A + v->param_off[6];
    const bool small = (1 + G) <= 32;
    const int GT = (G + 31) / 32, GK = 2 * GT;
    auto hid_col = [](int j, int kk) { return 16 * j + phi(kk); };
    auto in0_col = [&](int j, int kk) {            // cat([fts, PE(pts)]): natural feature k-steps then the PE arrangement
        if (j < KF) return 16 * j + kk;
        const int c = pe_src_col(L, 8 * (j - KF) + (kk & 7), kk >> 3);
        return c < 0 ? -1 : FT + c;
    };
    auto c0_col = [&](int j, int kk) {             // cat([h[...,1:], PE(dirs)]) voxnerf.py:248
        const int gk = small ? 1 : G / 16;
        if (j < gk) {
            const int row = 16 * j + phi(kk);      // index into the sigma-layer output tile(s)
            return small ? (row >= 1 && row <= G ? row - 1 : -1) : row;
        }
        const int c = pe_src_col(Lv, 8 * (j - gk) + (kk & 7), kk >> 3);
        return c < 0 ? -1 : G + c;
    };
    auto build = [&](StreamBuilder& sb) {
        sb.layer(sigma_w0, HD, d->input_ch, T, KF + PE_KS, false, in0_col);
        if (small) {
            sb.layer(sigma_w1, 1 + G, HD, 1, KS, false, hid_col);
        } else {
            sb.layer_rc(sigma_w1, HD, 1, KS, false, [](int, int r) { return r == 0 ? 0 : -1; }, hid_col);       // sigma row
            sb.layer_rc(sigma_w1, HD, G / 32, KS, false, [](int t, int r) { return 1 + 32 * t + r; }, hid_col); // geo rows
        }
        sb.layer(color_w0, HD, G + ICV, T, (small ? 1 : G / 16) + PEV_KS, false, c0_col);
        sb.layer(color_w1, HD, HD, T, KS, false, hid_col);
        sb.layer(color_w2, 3, HD, 1, KS, true, hid_col);
    };
    // the same network in the layer table of voxel_mlp_kernel.h VoxNet (any level): sigma and geo always separate layers, the
    // geo channels in GT zero-padded tiles
    auto build_pipe = [&](StreamBuilder& sb) {
        auto c0p = [&](int j, int kk) {
            if (j < GK) { const int c = 16 * j + phi(kk); return c < G ? c : -1; }
            const int c = pe_src_col(Lv, 8 * (j - GK) + (kk & 7), kk >> 3);
            return c < 0 ? -1 : G + c;
        };
        sb.layer(sigma_w0, HD, d->input_ch, T, KF + PE_KS, false, in0_col);
        sb.layer_rc(sigma_w1, HD, 1, KS, false, [](int, int r) { return r == 0 ? 0 : -1; }, hid_col);
        sb.layer_rc(sigma_w1, HD, GT, KS, false, [G](int t, int r) { return 32 * t + r < G ? 1 + 32 * t + r : -1; }, hid_col);
        sb.layer(color_w0, HD, G + ICV, T, GK + PEV_KS, false, c0p);
        sb.layer(color_w1, HD, HD, T, KS, false, hid_col);
        sb.layer(color_w2, 3, HD, 1, KS, true, hid_col);
    };
    for (int prec = 0; prec < EVD_VOX_NUM_PREC; ++prec) {
        StreamBuilder sb(prec);
        sb.arena = A;
        build(sb);
        v->nchunks[prec] = (int)(sb.bytes.size() / chunk_bytes(prec));
        rc = v->stream[prec].upload(sb);
        if (rc) { evd_voxel_destroy(v); return rc; }
        v->pipe_chunks[prec] = 0;
        v->train_chunks[prec] = 0;
        if (standard && voxel_pipe_built(prec, HD, G, FT)) {        // same layers, single-tile groups, 16 KiB chunks (voxel_mlp_kernel.h)
            StreamBuilder sp(prec, PIPE_CB);
            sp.arena = A;
            sp.group = 1;
            build_pipe(sp);
            v->pipe_chunks[prec] = (int)(sp.bytes.size() / PIPE_CB);
            rc = v->pipe[prec].upload(sp);
            if (rc) { evd_voxel_destroy(v); return rc; }
        }
        if (!is_train_prec(prec) || !standard) continue;
        {   // training: forward stream of the level (the coarse level has no pipelined inference kernel: its own copy) ...
            StreamBuilder sp(prec, PIPE_CB);
            sp.arena = A;
            sp.group = 1;
            build_pipe(sp);
            v->train_chunks[prec] = (int)(sp.bytes.size() / PIPE_CB);
            if ((rc = v->train[prec].upload(sp))) { evd_voxel_destroy(v); return rc; }
        }
        // ... and the W^T streams of the dgrad chain (voxel_train_kernel.h)
        auto put = [&](int which, auto fill) {
            StreamBuilder sb2(prec, PIPE_CB);
            sb2.arena = A;
            sb2.group = 1;
            fill(sb2);
            return v->bwd[prec][which].upload(sb2);
        };
        rc = put(VBWD_C2, [&](StreamBuilder& b) { b.layer_transposed(color_w2, 3, HD, 0, HD, nullptr, 0, T, 1, true, [](int, int kk) { return kk < 3 ? kk : -1; }); });
        if (!rc) rc = put(VBWD_C1, [&](StreamBuilder& b) { b.layer_transposed(color_w1, HD, HD, 0, HD, nullptr, 0, T, KS, true, hid_col); });
        // rows of an encoding appended to a W^T layer: output row idx of the extra tiles <-> encoding column, so that the output
        // fragments come out in the encoding's own arrangement (fragment j, position kk <-> pe_src_col(L, 8 j + (kk & 7), kk >> 3))
        auto enc_row = [](int L_, int idx) { const int j = idx / 16, kk = phi_inv(idx % 16); return pe_src_col(L_, 8 * j + (kk & 7), kk >> 3); };
        if (!rc) rc = put(VBWD_C0, [&](StreamBuilder& b) {       // [geo rows | direction-encoding rows] of color_net.0
            b.layer_at(GT + 1, KS, true,
                       [&](int t, int r) { if (t < GT) return 32 * t + r < G ? 32 * t + r : -1; const int c = enc_row(Lv, r); return c < 0 ? -1 : G + c; },
                       hid_col, [&](int r, int c) { return color_w0 + (size_t)c * (G + ICV) + r; });
        });
        if (!rc) rc = put(VBWD_SIGGEO, [&](StreamBuilder& b) {
            b.layer_transposed(sigma_w1, 1 + G, HD, 0, HD, nullptr, 0, T, GK + 1, true, [&](int j, int kk) {
                if (j < GK) { const int c = 16 * j + phi(kk); return c < G ? 1 + c : -1; }
                return kk == 0 ? 0 : -1;
            });
        });
        if (!rc) rc = put(VBWD_L0, [&](StreamBuilder& b) {       // [feature rows | point-encoding rows] of sigma_net.0
            const int FTT = (FT + 31) / 32, in_dim = d->input_ch;
            b.layer_at(FTT + 2, KS, true,
                       [&](int t, int r) { if (t < FTT) return 32 * t + r < FT ? 32 * t + r : -1; const int c = enc_row(L, 32 * (t - FTT) + r); return c < 0 ? -1 : FT + c; },
                       hid_col, [&](int r, int c) { return sigma_w0 + (size_t)c * in_dim + r; });
        });
        if (rc) { evd_voxel_destroy(v); return rc; }
    }
    if (standard && voxel_mlp_c_chunks(HD, G, FT) > 0) {     // compensated float16 mode: the layer table of voxel_mlp_c_kernel.h VoxNetC (groups of two tiles, 64-input blocks)
        auto c0p = [&](int j, int kk) {
            if (j < GK) { const int c = c_hid_col(j, kk); return c < G ? c : -1; }
            const int c = pe_src_col(Lv, 8 * (j - GK) + (kk & 7), kk >> 3);
            return c < 0 ? -1 : G + c;
        };
        auto sig_at = [=](int r, int c) -> const float* { return c < HD ? sigma_w1 + (size_t)r * HD + c : nullptr; };
        StreamBuilderC sc(PIPE_CB);
        sc.arena = A;
        sc.layer(sigma_w0, HD, d->input_ch, T, KF + PE_KS, 2, in0_col);
        sc.layer_at(1, KS, 1, [](int, int r) { return r == 0 ? 0 : -1; }, c_hid_col, sig_at);
        sc.layer_at(GT, KS, 2, [G](int t, int r) { return 32 * t + r < G ? 1 + 32 * t + r : -1; }, c_hid_col, sig_at);
        sc.layer(color_w0, HD, G + ICV, T, GK + PEV_KS, 2, c0p);
        sc.layer(color_w1, HD, HD, T, KS, 2, c_hid_col);
        sc.layer(color_w2, 3, HD, 1, KS, 1, c_hid_col);
        v->pipe_c_chunks = (int)(sc.bytes.size() / PIPE_CB);
        if (v->pipe_c_chunks != voxel_mlp_c_chunks(HD, G, FT))
            rc = fail(EVD_E_INVALID, "evd_voxel_create: f16c stream has %d chunks, kernel expects %d", v->pipe_c_chunks, voxel_mlp_c_chunks(HD, G, FT));
        if (!rc) rc = v->pipe_c.upload(sc);
        if (rc) { evd_voxel_destroy(v); return rc; }
    }
    {   // wgrad index maps (voxel_train.h)
        std::vector<int> m(VMAP_TOTAL, -1);
        for (int i = 0; i < 256; ++i) { const int c = hid_col(i / 16, i % 16); m[VMAP_HID + i] = c < HD ? c : -1; }
        for (int i = 0; i < 64; ++i) {
            m[VMAP_FTS + i] = i < FT ? i : -1;
            const int c = pe_src_col(L, 8 * (i / 16) + (i % 16 & 7), (i % 16) >> 3);
            m[VMAP_PE + i] = c < 0 ? -1 : FT + c;
        }
        for (int i = 0; i < 32; ++i) {
            const int c = pe_src_col(Lv, 8 * (i / 16) + (i % 16 & 7), (i % 16) >> 3);
            m[VMAP_DIR + i] = c < 0 ? -1 : G + c;
        }
        for (int i = 0; i < 128; ++i) {
            const int c = hid_col(i / 16, i % 16);
            m[VMAP_GEO_X + i] = c < G ? c : -1;
            m[VMAP_GEO_Y + i] = c < G ? 1 + c : -1;
        }
        for (int i = 0; i < 3; ++i) m[VMAP_COL + i] = i;
        m[VMAP_SIG] = 0;
        for (int i = 0; i < 16; ++i) { m[VMAP_F64_SG + i] = m[VMAP_GEO_Y + i]; m[VMAP_F64_SG + 16 + i] = m[VMAP_SIG + i]; }
        for (int i = 0; i < 32; ++i) { m[VMAP_F64_C0 + i] = m[VMAP_GEO_X + i]; m[VMAP_F64_C0 + 32 + i] = m[VMAP_DIR + i]; }
        for (int i = 0; i < 32; ++i) m[VMAP_F64_L0 + i] = m[VMAP_FTS + i];
        for (int i = 0; i < 64; ++i) m[VMAP_F64_L0 + 32 + i] = m[VMAP_PE + i];
        for (int i = 0; i < 128; ++i) m[VMAP_SG5 + i] = m[VMAP_GEO_Y + i];
        for (int i = 0; i < 32; ++i) m[VMAP_SG5 + 128 + i] = m[VMAP_SIG + i];
        if ((rc = v->wmaps.upload(m.data(), m.size() * sizeof(int)))) { evd_voxel_destroy(v); return rc; }
    }
    std::vector<float> b(32 * 16, 0.f);            // zero block shared by the bias-free sigma layers
    std::vector<int32_t> bsrc(32 * 16, -1);
    auto push = [&](int block, int out_dim, int tiles) {
        for (int i = 0; i < tiles * 32; ++i) {
            b.push_back(i < out_dim ? A[v->param_off[block] + i] : 0.f);
            bsrc.push_back(i < out_dim ? (int32_t)(v->param_off[block] + i) : -1);
        }
    };
    push(3, HD, T);
    push(5, HD, T);
    push(7, 3, 1);
    rc = v->bias.upload(b.data(), b.size() * sizeof(float));
    if (!rc) rc = v->bias_src.upload(bsrc.data(), bsrc.size() * sizeof(int32_t));
    if (rc) { evd_voxel_destroy(v); return rc; }
    *out = v;
    return EVD_OK;
}

int evd_voxel_sample(const evd_voxel* v, const float* pts, long n, float* out, int out_stride, int out_col, void* stream) {
    EVD_REQUIRE(v && out && n >= 0 && out_stride >= out_col + v->app_dim, "evd_voxel_sample: bad arguments");
    if (n == 0) return EVD_OK;
    return launch_voxel_sample(v->gp, false, pts, n, out, out_stride, out_col, as_stream(stream));
}

// sampling inside the c2f render: the half-precision arithmetic modes read the float16 copies of the grids
// feeds_only: the features go to the NEXT level's networks and to nothing that places samples (-1: decide by the level -- a level fed
// by the previous one is the last of a c2f render)
static bool grids_half_for(const evd_voxel* v, int precision, int feeds_only = -1) {
    static const bool f32_grids = env_flag("EVD_F32_GRIDS");   // developer switch: float32 grids in every mode
    bool half = (precision == EVD_PREC_BF16 || precision == EVD_PREC_F16) && !f32_grids;
    if (precision == EVD_PREC_F16C && !f32_grids) {
        // The compensated mode: the level that is FED by the previous one (its features are cat([previous level's, its own]) -- the fine
        // level of a c2f render, whose output is the image) reads the float16 copies of its grids: their 2^-12 rounding is random over 96
        // channels and six taps and costs nothing measurable (trained blurfactory-size model: RGB L-inf 8.4e-6 against 7.9e-6 on the
        // float32 grids, bound 1e-4), and the gather moves half the bytes (c2f render 0.708 -> 0.683 ms).  The coarse level keeps the
        // float32 grids: its weights place the importance samples, and with float16 grids there the FINE image is at 1.0e-4.
        // The coarse features AT THE IMPORTANCE SAMPLES (renderer.py:209) also feed the fine networks only (bit 2) -- but there the
        // float16 grids cost 3.4e-5 of the fine image for ~1 % of the render: left on the float32 grids.
        // EVD_F16C_HALF_GRIDS (developer switch, read per call): bit 0 the coarse level's first gather, bit 1 the fine level, bit 2 the
        // coarse level's gather at the importance samples; default 2.
        const char* e = getenv("EVD_F16C_HALF_GRIDS");
        const int m = e ? atoi(e) : 2;
        const bool last = v->ft_dim > v->app_dim;
        half = (m & (last ? 2 : (feeds_only == 1 ? 4 : 1))) != 0;
    }
    return half;
}
static int sample_for(const evd_voxel* v, int precision, const float* pts, long n, float* out, int out_stride, int out_col, void* stream, int feeds_only = -1) {
    return launch_voxel_sample(v->gp, grids_half_for(v, precision, feeds_only), pts, n, out, out_stride, out_col, as_stream(stream));
}

int evd_voxel_sample_prec(const evd_voxel* v, int precision, const float* pts, long n, float* out, int out_stride, int out_col, void* stream) {
    EVD_REQUIRE(v && out && n >= 0 && out_stride >= out_col + v->app_dim, "evd_voxel_sample_prec: bad arguments");
    EVD_REQUIRE(precision >= 0 && precision <= EVD_PREC_F16M, "evd_voxel_sample_prec: unknown precision %d", precision);     // (F16M: the float32 grids)
    if (n == 0) return EVD_OK;
    return sample_for(v, precision, pts, n, out, out_stride, out_col, stream);
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// composite_feature levels: the per-sample geo rows [R S, G] and the composited map [R, G]
static size_t composite_scratch_bytes(const evd_voxel* v, long R, int S) {
    return v->composite_feature ? align256((size_t)R * S * v->geo * 4) + align256((size_t)R * v->geo * 4) : 0;
}

size_t evd_voxel_forward_workspace_bytes(const evd_voxel* v, long R, int S) {
    if (!v || R < 0 || S < 1) return 0;
    return align256((size_t)R * S * 16) + composite_scratch_bytes(v, R, S) + 512;
}

// activation of the geo rows before they are composited (voxnerf.py:172: rgb_activate(raw[..., 1:]))
static __global__ __launch_bounds__(256) void k_act_inplace(float* __restrict__ x, long n, int code) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] = act(code, x[i]);
}

// colour network of a composite_feature level, per RAY (voxnerf.py:231-239): h = cat([feature_map, PE(dirs)]) -> Linear + ReLU
// (num_layers_color - 1 times) -> Linear -> sigmoid.  A wavefront per ray, a lane per hidden unit (HD <= 256: up to 4 units per lane);
// the weights are the reference's float32 [out, in] rows, read through the caches (7 k MACs per ray: nothing to optimise).
static __global__ __launch_bounds__(256) void k_color_rows(const float* __restrict__ cnet, const float* __restrict__ fm, const float* __restrict__ viewdirs,
                                                           int vd_stride, long R, int G, int HD, int Lv, float* __restrict__ color) {
    __shared__ float in[4][160], h0[4][256], h1[4][256];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + w;
    const int icv = 3 * (1 + 2 * Lv), nin = G + icv;
    const float *w0 = cnet, *b0 = w0 + (long)HD * nin, *w1 = b0 + HD, *b1 = w1 + (long)HD * HD, *w2 = b1 + HD, *b2 = w2 + 3L * HD;
    if (r < R) {
        for (int i = lane; i < nin; i += 64) {
            float x;
            if (i < G) x = fm[r * G + i];
            else {
                const int c = i - G;              // embedding.py:88-98: [x, sin(x f0), cos(x f0), sin(x f1), ...], 3 values each
                const float d = viewdirs[r * vd_stride + c % 3];
                x = c < 3 ? d : ((c - 3) / 3 % 2 == 0 ? sinf(d * (float)(1 << ((c - 3) / 6))) : cosf(d * (float)(1 << ((c - 3) / 6))));
            }
            in[w][i] = x;
        }
    }
    __syncthreads();
    if (r < R)
        for (int j = lane; j < HD; j += 64) {
            float s = b0[j];
            for (int i = 0; i < nin; ++i) s = fmaf(w0[(long)j * nin + i], in[w][i], s);
            h0[w][j] = fmaxf(s, 0.f);
        }
    __syncthreads();
    if (r < R)
        for (int j = lane; j < HD; j += 64) {
            float s = b1[j];
            for (int i = 0; i < HD; ++i) s = fmaf(w1[(long)j * HD + i], h0[w][i], s);
            h1[w][j] = fmaxf(s, 0.f);
        }
    __syncthreads();
    if (r < R && lane < 3) {
        float s = b2[lane];
        for (int i = 0; i < HD; ++i) s = fmaf(w2[(long)lane * HD + i], h1[w][i], s);
        color[r * 3 + lane] = 1.f / (1.f + expf(-s));            // torch.sigmoid(h) voxnerf.py:239
    }
}

static int voxel_pass(const evd_voxel* v, int precision, const float* pts, const float* viewdirs, int vd_stride, const float* fts,
                      int ft_stride, const float* z, const float* rays_d, int rd_stride, long R, int S, int is_train,
                      const float* noise, float* color, float* depth, float* acc, float* weights, float* feature, float* raw,
                      void* stream, float* cscratch = nullptr) {
    // composite_feature (PBE, voxnerf.py:223-239): the per-sample geo rows are composited, the colour network runs per ray; `feature`
    // is then the composited map [R, G] (NULL: not wanted) and cscratch holds the rows + the map
    float* crows = nullptr;
    float* cmap = nullptr;
    if (v->composite_feature) {
        if (!cscratch) return fail(EVD_E_WORKSPACE, "evd_voxel: a composite_feature level needs its scratch (workspace sized by the *_workspace_bytes query of this handle)");
        crows = cscratch;
        cmap = feature ? feature : (float*)((char*)cscratch + align256((size_t)R * S * v->geo * 4));
        feature = crows;
    }
    VoxMlpParams p;
    static const bool no_pipe = env_flag("EVD_NO_PIPE");
    const bool comp = precision == EVD_PREC_F16C && v->pipe_c_chunks > 0;
    const bool coarse_of_f16c = precision == EVD_PREC_F16C && !comp;
    if (coarse_of_f16c) precision = EVD_PREC_F16X3;      // the 64-wide coarse level: float32-grade arithmetic (a few % of the render)
    if (comp && feature) return fail(EVD_E_INVALID, "evd_voxel: per-sample feature rows are not built in EVD_PREC_F16C (use EVD_PREC_F16X3)");
    const bool piped = v->pipe_chunks[precision] > 0 && !no_pipe;
    p.wstream = (const char*)(piped ? v->pipe[precision].data.p : v->stream[precision].data.p);
    p.bias = (const float*)v->bias.p;
    p.pts = pts; p.viewdirs = viewdirs; p.fts = fts; p.nsamp = R * (long)S; p.S = S; p.vd_stride = vd_stride; p.ft_stride = ft_stride;
    p.nchunks = piped ? v->pipe_chunks[precision] : v->nchunks[precision]; p.nbias = (int)(v->bias.bytes / sizeof(float)); p.raw = raw; p.feature = feature; p.act = nullptr;
    p.pe_l = v->multires; p.pe_lv = v->multires_views;
    if (comp) {
        if (v->pipe_c_stale) {
            int rcc = repack_stream_c(v->pipe_c, (const float*)v->arena_dev.p, as_stream(stream));
            if (rcc) return rcc;
            v->pipe_c_stale = false;
        }
        p.wstream = (const char*)v->pipe_c.data.p;
        p.wscale = (const unsigned*)v->pipe_c.scales.p;
        p.nchunks = v->pipe_c_chunks;
    }
    // the 64-wide coarse level on the software pipeline (its training forward's stream and layer table), where that is built
    const bool coarse_pipe = !comp && !piped && !no_pipe && !feature && v->hidden_dim == 64 && v->geo == 15 && v->ft_dim == 32 &&
                             is_train_prec(precision) && v->train_chunks[precision] > 0;
    if (coarse_pipe) {
        p.wstream = (const char*)v->train[precision].data.p;
        p.nchunks = v->train_chunks[precision];
        p.rev_trig = coarse_of_f16c ? 1 : 0;       // an f16c render: this level's encodings as the fine level's kernel computes them (voxel_mlp_kernel.h)
    }
    int rc = comp ? launch_voxel_pipe_f16c(p, as_stream(stream))
             : coarse_pipe ? (precision == EVD_PREC_BF16 ? launch_voxel_coarse_pipe_bf16(p, as_stream(stream))
                              : precision == EVD_PREC_F16 ? launch_voxel_coarse_pipe_f16(p, as_stream(stream)) : launch_voxel_coarse_pipe_f16x3(p, as_stream(stream)))
             : piped ? (precision == EVD_PREC_BF16 ? launch_voxel_pipe_bf16(feature != nullptr, p, as_stream(stream))
                      : precision == EVD_PREC_F16 ? launch_voxel_pipe_f16(feature != nullptr, p, as_stream(stream))
                                                  : launch_voxel_pipe_f16x3(feature != nullptr, p, as_stream(stream)))
                   : voxel_mlp_dispatch(precision, v->hidden_dim, v->geo, v->ft_dim, p, as_stream(stream));
    if (rc) return rc;
    const float thr = (!is_train && v->rmnear > 0.f) ? (float)((double)v->rmnear / 128.0) : 0.f;
    if (v->composite_feature) {
        hipStream_t st = as_stream(stream);
        const long ng = R * (long)S * v->geo;
        if (v->rgb_act != EVD_ACT_NONE) {
            hipLaunchKernelGGL(k_act_inplace, dim3((unsigned)cdiv(ng, 256L)), dim3(256), 0, st, crows, ng, v->rgb_act);
            EVD_LAUNCH_CHECK();
        }
        // weights / depth / acc from the densities; the geo rows ride along as the feature map (sum_s w feature); the per-sample colour
        // composite written to `color` here is overwritten by the per-ray colour network below
        if ((rc = evd_raw2outputs(raw, z, rays_d, rd_stride, R, S, 4, 0, 1, 3, v->rgb_act, v->sigma_act, 0, thr, noise,
                                  color, nullptr, acc, weights, depth, crows, v->geo, cmap, stream))) return rc;
        hipLaunchKernelGGL(k_color_rows, dim3((unsigned)cdiv(R, 4L)), dim3(256), 0, st, (const float*)v->cnet.p, cmap, viewdirs, vd_stride, R, v->geo,
                           v->hidden_dim, v->multires_views, color);
        EVD_LAUNCH_CHECK();
        return EVD_OK;
    }
    return evd_raw2outputs(raw, z, rays_d, rd_stride, R, S, 4, 0, 1, 3, v->rgb_act, v->sigma_act, 0, thr, noise,
                           color, nullptr, acc, weights, depth, nullptr, 0, nullptr, stream);
}

int evd_voxel_forward(const evd_voxel* v, int precision, const float* pts, const float* viewdirs, int vd_stride, const float* fts, int F,
                      const float* z, const float* rays_d, int rays_d_stride, long R, int S, int is_train,
                      float* color, float* depth, float* acc, float* weights, float* feature,
                      void* workspace, size_t workspace_bytes, void* stream) {
    EVD_REQUIRE(v && pts && viewdirs && fts && z && rays_d, "evd_voxel_forward: null argument");
    EVD_REQUIRE(precision >= 0 && precision <= EVD_PREC_F16C, "evd_voxel_forward: unknown precision %d", precision);
    EVD_REQUIRE(F == v->ft_dim, "evd_voxel_forward: fts has %d channels, this level takes %d", F, v->ft_dim);
    EVD_REQUIRE(weights, "evd_voxel_forward: the weights output is required");
    if (R == 0) return EVD_OK;
    const size_t need = evd_voxel_forward_workspace_bytes(v, R, S);
    if (!workspace || workspace_bytes < need) return fail(EVD_E_WORKSPACE, "evd_voxel_forward: workspace %zu < %zu bytes", workspace_bytes, need);
    float* raw = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    float* cs = v->composite_feature ? (float*)((char*)raw + align256((size_t)R * S * 16)) : nullptr;
    return voxel_pass(v, precision, pts, viewdirs, vd_stride, fts, F, z, rays_d, rays_d_stride, R, S, is_train, nullptr,
                      color, depth, acc, weights, feature, raw, stream, cs);
}

size_t evd_c2f_render_workspace_bytes(const evd_voxel* coarse, const evd_voxel* fine, const evd_render_cfg* cfg, long R) {
    if (!coarse || !cfg || R < 0) return 0;
    const size_t S = cfg->N_samples, Ni = cfg->N_importance > 0 ? cfg->N_importance : 0, St = S + Ni, r = (size_t)R;
    size_t b = 0;
    b += align256(r * 11 * 4);              // ray_batch
    b += align256(r * S * 4);               // z0
    b += align256(r * St * 4);              // z merged
    b += align256(r * St * 12);             // pts
    b += align256(r * St * 64 * 4);         // features [n,64]
    b += align256(r * S * 32 * 4);          // coarse features at the coarse samples [R*S,32]
    b += align256(r * (Ni ? Ni : 1) * 32 * 4);   // coarse features at the new samples
    b += align256(r * (Ni ? Ni : 1) * 12);  // new points
    b += align256(r * St * 4);              // sort order
    b += align256(r * St * 16);             // raw
    b += align256(r * S * 4);               // weights0
    b += align256(r * St * 4);              // weights
    b += align256(r * (Ni ? Ni : 1) * 4);   // z_samples
    b += composite_scratch_bytes(coarse, R, (int)S);       // a composite_feature coarse level (kernel_type PBE): geo rows + composited map
    if (fine && Ni) b += composite_scratch_bytes(fine, R, (int)St);
    return b + 512;
}

int evd_c2f_render_rays(const evd_voxel* coarse, const evd_voxel* fine, const evd_render_cfg* cfg, const float* rb, long R,
                        const float* t_rand, const float* u, const float* noise0, const float* noise1,
                        evd_render_out* out, void* workspace, size_t workspace_bytes, void* stream) {
    EVD_REQUIRE(coarse && cfg && out && (rb || R == 0), "evd_c2f_render_rays: null argument");
    EVD_REQUIRE(cfg->use_viewdirs, "evd_c2f_render_rays: use_viewdirs=False is not supported");
    EVD_REQUIRE(cfg->N_importance <= 0 || fine, "evd_c2f_render_rays: N_importance > 0 needs the fine level");
    EVD_REQUIRE(cfg->N_importance <= 0 || cfg->N_samples >= 3, "evd_c2f_render_rays: hierarchical sampling needs N_samples >= 3");
    EVD_REQUIRE(!(cfg->perturb > 0.f) || (t_rand && (cfg->N_importance <= 0 || u)), "evd_c2f_render_rays: perturb > 0 needs explicit draws");
    EVD_REQUIRE(coarse->ft_dim == coarse->app_dim && (!fine || fine->ft_dim == coarse->app_dim + fine->app_dim),
                "evd_c2f_render_rays: feature widths do not chain (coarse app_dim %d, fine input %d)", coarse->app_dim, fine ? fine->ft_dim : -1);
    if (R == 0) return EVD_OK;
    const size_t need = evd_c2f_render_workspace_bytes(coarse, fine, cfg, R);
    if (!workspace || workspace_bytes < need) return fail(EVD_E_WORKSPACE, "evd_c2f_render_rays: workspace %zu < %zu bytes", workspace_bytes, need);
    const int S = cfg->N_samples, Ni = cfg->N_importance > 0 ? cfg->N_importance : 0, St = S + Ni;
    hipStream_t st = as_stream(stream);
    char* w = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    auto take = [&](size_t bytes) { char* p = w; w += align256(bytes); return (float*)p; };
    const size_t r = (size_t)R;
    (void)take(r * 11 * 4);
    float* z0 = take(r * S * 4);
    float* z2 = take(r * St * 4);
    float* pts = take(r * St * 12);
    float* ft = take(r * St * 64 * 4);
    float* ft0 = take(r * S * 32 * 4);
    float* ftn = take(r * (Ni ? Ni : 1) * 32 * 4);
    float* ptn = take(r * (Ni ? Ni : 1) * 12);
    int* order = (int*)take(r * St * 4);
    float* raw = take(r * St * 16);
    float* wts0 = take(r * S * 4);
    float* wts = take(r * St * 4);
    float* zs = take(r * (Ni ? Ni : 1) * 4);
    float* cs0 = coarse->composite_feature ? (float*)take(composite_scratch_bytes(coarse, R, S)) : nullptr;
    float* cs1 = (fine && Ni && fine->composite_feature) ? (float*)take(composite_scratch_bytes(fine, R, St)) : nullptr;
    const int FS = 64;      // feature row stride: coarse features at column 0, fine at column coarse->app_dim (renderer.py:195)
    int rc;
    float* zc = (Ni ? (out->z_vals0 ? out->z_vals0 : z0) : (out->z_vals ? out->z_vals : z0));
    if ((rc = launch_sample_z_pts(cfg, rb, 11, R, t_rand, zc, pts, st))) return rc;       // z stratification + pts = o + d z (renderer.py:163-180)
    const int FC = coarse->app_dim;
    EVD_REQUIRE(!Ni || (FC == 32 && FC % 4 == 0), "evd_c2f_render_rays: coarse app_dim %d (workspace is sized for 32)", FC);
    EVD_REQUIRE(FC <= FS && (!Ni || !fine || FC + fine->app_dim <= FS),
                "evd_c2f_render_rays: app_dim %d + %d exceeds the feature row stride %d", FC, (Ni && fine) ? fine->app_dim : 0, FS);
    if (!Ni) {
        if ((rc = sample_for(coarse, cfg->precision, pts, R * (long)S, ft, FS, 0, stream))) return rc;      // renderer.py:183
        float* wo = out->weights ? out->weights : wts;
        return voxel_pass(coarse, cfg->precision, pts, rb + 8, 11, ft, FS, zc, rb + 3, 11, R, S, cfg->is_train, noise0,
                          out->rgb, out->depth, out->acc, wo, out->feature, out->raw ? out->raw : raw, stream, cs0);
    }
    if ((rc = sample_for(coarse, cfg->precision, pts, R * (long)S, ft0, FC, 0, stream))) return rc;          // renderer.py:183
    float* w0 = out->weights0 ? out->weights0 : wts0;
    if ((rc = voxel_pass(coarse, cfg->precision, pts, rb + 8, 11, ft0, FC, zc, rb + 3, 11, R, S, cfg->is_train, noise0,
                         out->rgb0, out->depth0, out->acc0, w0, nullptr, raw, stream, cs0))) return rc;
    float* zm = out->z_vals ? out->z_vals : z2;
    // (+ the positions of the new and of the merged samples: the coarse pass is done with `pts`)
    if ((rc = launch_sample_pdf_merge_pts(zc, w0, R, S, Ni, cfg->perturb == 0.f, u, zs, zm, order, out->z_std, rb, 11, ptn, pts, st))) return rc;
    // merged sample set (renderer.py:205-213), as the reference does it: coarse features are sampled at the NEW points only
    // (:209) and the rows of the old and new points are gathered by the sort order (:212-213); the fine level is sampled at
    // all merged points (:211 samples the new ones and :194 the old ones -- a pure function of the point either way).
    const long n2 = R * (long)St;
    if ((rc = sample_for(coarse, cfg->precision, ptn, R * (long)Ni, ftn, FC, 0, stream, 1))) return rc;
    if ((rc = launch_merge_features(ft0, ftn, order, R, S, Ni, FC, ft, FS, st))) return rc;
    if ((rc = sample_for(fine, cfg->precision, pts, n2, ft, FS, coarse->app_dim, stream))) return rc;
    float* wo = out->weights ? out->weights : wts;
    return voxel_pass(fine, cfg->precision, pts, rb + 8, 11, ft, FS, zm, rb + 3, 11, R, St, cfg->is_train, noise1,
                      out->rgb, out->depth, out->acc, wo, out->feature, out->raw ? out->raw : raw, stream, cs1);
}

int evd_c2f_render(const evd_voxel* coarse, const evd_voxel* fine, const evd_render_cfg* cfg, const float* rays, long R,
                   const float* t_rand, const float* u, const float* noise0, const float* noise1,
                   evd_render_out* out, void* workspace, size_t workspace_bytes, void* stream) {
    EVD_REQUIRE(cfg && coarse && (rays || R == 0), "evd_c2f_render: null argument");
    if (R == 0) return EVD_OK;
    const size_t need = evd_c2f_render_workspace_bytes(coarse, fine, cfg, R);
    if (!workspace || workspace_bytes < need) return fail(EVD_E_WORKSPACE, "evd_c2f_render: workspace %zu < %zu bytes", workspace_bytes, need);
    float* rb = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    int rc = evd_ray_batch(cfg, rays, R, rb, stream);
    if (rc) return rc;
    return evd_c2f_render_rays(coarse, fine, cfg, rb, R, t_rand, u, noise0, noise1, out, workspace, workspace_bytes, stream);
}

int evd_points(const float* ray_batch, int ncol, const float* z, long R, int S, float* pts, void* stream) {
    EVD_REQUIRE(ray_batch && z && pts && ncol >= 6 && R >= 0 && S >= 1, "evd_points: bad arguments");
    if (R == 0) return EVD_OK;
    return launch_points(ray_batch, ncol, z, R * (long)S, S, pts, as_stream(stream));
}

// ---- the coarse feature rows of the merged sample set (renderer.py:209-213), as an operation of its own for the training path
int evd_merge_features(const float* old, const float* fresh, const int* order, long R, int S, int N, int F, float* out, int out_stride, void* stream) {
    EVD_REQUIRE(old && fresh && order && out && R >= 0 && S >= 1 && N >= 1 && F >= 4 && F % 4 == 0 && out_stride >= F && out_stride % 4 == 0,
                "evd_merge_features: bad arguments (F and out_stride multiples of 4)");
    if (R == 0) return EVD_OK;
    return launch_merge_features(old, fresh, order, R, S, N, F, out, out_stride, as_stream(stream));
}

int evd_merge_features_bwd(const float* d_out, int d_stride, const int* order, long R, int S, int N, int F, float* d_old, float* d_fresh, void* stream) {
    EVD_REQUIRE(d_out && order && d_old && d_fresh && R >= 0 && S >= 1 && N >= 1 && F >= 4 && F % 4 == 0 && d_stride >= F && d_stride % 4 == 0,
                "evd_merge_features_bwd: bad arguments (F and d_stride multiples of 4)");
    if (R == 0) return EVD_OK;
    return launch_merge_features_bwd(d_out, d_stride, order, R, S, N, F, d_old, d_fresh, as_stream(stream));
}

// ---- training: the level's sigma / colour networks -----------------------------------------------------
static const int VOX_WGRAD_BLOCKS = 256;
static long vox_tiles(long nsamp) { return cdiv(nsamp, 256L) * 8; }
// EVD_PREC_F16C trains on the single-product float16 store and backward behind a compensated forward: the level's f16c kernel where
// that is built (fine level), else the split-float16 forward writing the float16 store (coarse level)
// EVD_PREC_F16M: the split-float16 forward on every level, same store and backward
static bool vox_train_built(const evd_voxel* v, int prec) {
    if (prec == EVD_PREC_F16C) return v->train_chunks[EVD_PREC_F16] > 0 && (v->pipe_c_chunks > 0 || v->train_chunks[EVD_PREC_F16X3] > 0);
    if (prec == EVD_PREC_F16M) return v->train_chunks[EVD_PREC_F16] > 0 && v->train_chunks[EVD_PREC_F16X3] > 0;
    return prec >= 0 && prec < EVD_VOX_NUM_PREC && is_train_prec(prec) && v->train_chunks[prec] > 0;
}
static int vox_store_prec(int prec) { return (prec == EVD_PREC_F16C || prec == EVD_PREC_F16M) ? EVD_PREC_F16 : prec; }

long evd_voxel_param_count(const evd_voxel* v) { return v ? v->param_off[8] : 0; }

int evd_voxel_param_blocks(const evd_voxel* v, long* offsets, int capacity) {
    EVD_REQUIRE(v, "evd_voxel_param_blocks: null level");
    if (offsets)
        for (int i = 0; i <= 8 && i < capacity; ++i) offsets[i] = v->param_off[i];
    return 8;
}

int evd_voxel_load_params(evd_voxel* v, const float* params, void* stream) {
    EVD_REQUIRE(v && params, "evd_voxel_load_params: null argument");
    hipStream_t st = as_stream(stream);
    int rc;
    std::vector<PackedStream*> all;
    for (int i = 0; i < EVD_VOX_NUM_PREC; ++i) {
        all.push_back(&v->stream[i]);
        all.push_back(&v->pipe[i]);
        all.push_back(&v->train[i]);
        for (int k = 0; k < VBWD_NSTREAMS; ++k) all.push_back(&v->bwd[i][k]);
    }
    if ((rc = repack_batch(v->batch, all, params, st))) return rc;
    if (v->pipe_c_chunks > 0) {
        const size_t bytes = (size_t)v->param_off[8] * sizeof(float);
        if (!v->arena_dev.p && (rc = v->arena_dev.alloc(bytes))) return rc;
        EVD_HIP(hipMemcpyAsync(v->arena_dev.p, params, bytes, hipMemcpyDeviceToDevice, st));
        v->pipe_c_stale = true;
    }
    if (v->cnet.p) EVD_HIP(hipMemcpyAsync(v->cnet.p, params + v->param_off[2], v->cnet.bytes, hipMemcpyDeviceToDevice, st));
    const long nb = (long)(v->bias.bytes / sizeof(float));
    hipLaunchKernelGGL(k_gather_f32, dim3((unsigned)cdiv(nb, 256L)), dim3(256), 0, st, params, (const int*)v->bias_src.p, nb, (float*)v->bias.p);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

size_t evd_voxel_train_store_bytes(const evd_voxel* v, long nsamp) {
    return (!v || nsamp < 0) ? 0 : (size_t)vox_tiles(nsamp) * voxel_store_tile_bytes(v->hidden_dim);
}
size_t evd_voxel_train_store_bytes_prec(const evd_voxel* v, int precision, long nsamp) {
    return (!v || nsamp < 0 || !vox_train_built(v, precision)) ? 0
           : (size_t)vox_tiles(nsamp) * voxel_store_tile_bytes_prec(v->hidden_dim, vox_store_prec(precision));
}

size_t evd_voxel_backward_workspace_bytes(void) { return (size_t)VOX_WGRAD_BLOCKS * 8 * 9 * 4096 + 512; }

int evd_voxel_mlp_train(const evd_voxel* v, int precision, const float* pts, const float* viewdirs, int vd_stride, const float* fts, int ft_stride,
                        long R, int S, float* raw, float* feature, void* store, size_t store_bytes, void* stream) {
    EVD_REQUIRE(v && pts && viewdirs && fts && raw && store, "evd_voxel_mlp_train: null argument");
    EVD_REQUIRE(!v->composite_feature, "evd_voxel_mlp_train: composite_feature levels (kernel_type PBE) are built for inference only");
    EVD_REQUIRE(vox_train_built(v, precision), "evd_voxel_mlp_train: the training path is built for precision f16 / bf16 / f16x3 / f16c / f16m");
    EVD_REQUIRE(R >= 0 && S >= 1 && ft_stride >= v->ft_dim && ft_stride % 4 == 0, "evd_voxel_mlp_train: bad shape");
    if (R == 0) return EVD_OK;
    const long nsamp = R * (long)S;
    if (store_bytes < evd_voxel_train_store_bytes_prec(v, precision, nsamp))
        return fail(EVD_E_WORKSPACE, "evd_voxel_mlp_train: store %zu < %zu bytes", store_bytes, evd_voxel_train_store_bytes_prec(v, precision, nsamp));
    VoxMlpParams p;
    p.bias = (const float*)v->bias.p;
    p.pts = pts; p.viewdirs = viewdirs; p.fts = fts; p.nsamp = nsamp; p.S = S; p.vd_stride = vd_stride; p.ft_stride = ft_stride;
    p.nbias = (int)(v->bias.bytes / sizeof(float)); p.raw = raw; p.feature = feature; p.act = (char*)store;
    if (precision == EVD_PREC_F16C || precision == EVD_PREC_F16M) {
        if (feature) return fail(EVD_E_INVALID, "evd_voxel_mlp_train: float32 feature rows are not built in EVD_PREC_F16C / EVD_PREC_F16M (the geo features stay fragments in the store)");
        if (precision == EVD_PREC_F16C && v->pipe_c_chunks > 0) {        // the level's compensated kernel (fine level)
            if (v->pipe_c_stale) {
                int rcc = repack_stream_c(v->pipe_c, (const float*)v->arena_dev.p, as_stream(stream));
                if (rcc) return rcc;
                v->pipe_c_stale = false;
            }
            p.wstream = (const char*)v->pipe_c.data.p;
            p.wscale = (const unsigned*)v->pipe_c.scales.p;
            p.nchunks = v->pipe_c_chunks;
            return launch_voxel_train_fwd_f16c(p, as_stream(stream));
        }
        p.wstream = (const char*)v->train[EVD_PREC_F16X3].data.p;
        p.nchunks = v->train_chunks[EVD_PREC_F16X3];
        return launch_voxel_train_fwd_f16x3_hi(v->hidden_dim, p, as_stream(stream));
    }
    p.wstream = (const char*)v->train[precision].data.p;
    p.nchunks = v->train_chunks[precision];
    return launch_voxel_train_fwd_dispatch(precision, v->hidden_dim, p, as_stream(stream));
}

int evd_voxel_geo_feat_dim(const evd_voxel* v) { return v ? v->geo : 0; }

int evd_voxel_mlp_backward(const evd_voxel* v, int precision, const float* d_raw, const float* raw, const float* d_feature, const void* awp_store,
                           size_t awp_store_bytes, long R, int S, void* store,
                           size_t store_bytes, const evd_voxel_grads* grads, float* d_fts, int d_fts_stride, const float* pts, const float* viewdirs, int vd_stride,
                           float* d_pts, float* d_dirs, void* workspace, size_t workspace_bytes, void* stream) {
    EVD_REQUIRE((!d_pts || pts) && (!d_dirs || viewdirs), "evd_voxel_mlp_backward: d_pts / d_dirs need the forward's pts / viewdirs");
    EVD_REQUIRE(v && d_raw && raw && store && grads && workspace, "evd_voxel_mlp_backward: null argument");
    EVD_REQUIRE(vox_train_built(v, precision), "evd_voxel_mlp_backward: the training path is built for precision f16 / bf16 / f16x3 / f16c / f16m");
    EVD_REQUIRE(R >= 0 && S >= 1 && (!d_fts || d_fts_stride >= v->ft_dim), "evd_voxel_mlp_backward: bad shape");
    if (R == 0) return EVD_OK;
    const long nsamp = R * (long)S;
    if (store_bytes < evd_voxel_train_store_bytes_prec(v, precision, nsamp))
        return fail(EVD_E_WORKSPACE, "evd_voxel_mlp_backward: store %zu < %zu bytes", store_bytes, evd_voxel_train_store_bytes_prec(v, precision, nsamp));
    if (workspace_bytes < evd_voxel_backward_workspace_bytes()) return fail(EVD_E_WORKSPACE, "evd_voxel_mlp_backward: workspace %zu < %zu bytes", workspace_bytes, evd_voxel_backward_workspace_bytes());
    precision = vox_store_prec(precision);      // EVD_PREC_F16C: the float16 mode's store, W^T streams and kernels
    VoxBwdPlan b;
    b.d_raw = d_raw; b.raw = raw; b.d_feature = d_feature; b.nsamp = nsamp; b.tiles = vox_tiles(nsamp); b.store = (char*)store;
    b.awp_store = nullptr; b.awp_tile_bytes = 0; b.awp_slot = 0; b.awp_words = nullptr;
    if (awp_store) {            // the fused AWP embedding ran its backward on this sample set: its d geo fragments join this level's
        const size_t need = (size_t)awp_tiles(nsamp) * awpstore::TILE_BYTES + awpstore::TRAILER_BYTES;
        EVD_REQUIRE(v->geo == AWP_IN, "evd_voxel_mlp_backward: awp_store goes with the fine level (geo %d)", AWP_IN);
        EVD_REQUIRE(is_half_prec(precision), "evd_voxel_mlp_backward: awp_store goes with the f16 / bf16 modes (in f16x3 pass the geo gradient as d_feature rows)");
        if (awp_store_bytes < need) return fail(EVD_E_WORKSPACE, "evd_voxel_mlp_backward: awp_store %zu < %zu bytes", awp_store_bytes, need);
        b.awp_store = (const char*)awp_store; b.awp_tile_bytes = awpstore::TILE_BYTES; b.awp_slot = awpstore::D_GEO;
        b.awp_words = (const unsigned*)(b.awp_store + awp_tiles(nsamp) * awpstore::TILE_BYTES);
    }
    for (int k = 0; k < VBWD_NSTREAMS; ++k) b.wt[k] = (const char*)v->bwd[precision][k].data.p;
    b.maps = (const int*)v->wmaps.p;
    char* w = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    b.maxbits = (unsigned*)w;
    b.partial = (float*)(w + 256);
    b.wgrad_blocks = VOX_WGRAD_BLOCKS;
    b.side = nullptr; b.ev = nullptr;
    static const bool overlap = env_flag("EVD_BWD_OVERLAP_VOXEL");     // measured: no gain for the level networks (7.70 vs 7.86 ms per c2f iteration), off
    if (overlap) {
        if (const int nb = bwd_overlap_blocks(VOX_WGRAD_BLOCKS)) {
            int rc0 = v->side.get(&b.side, &b.ev);
            if (rc0) return rc0;
            b.wgrad_blocks = nb;
        }
    }
    b.d_fts = d_fts; b.d_fts_stride = d_fts_stride;
    b.pts = pts; b.viewdirs = viewdirs; b.vd_stride = vd_stride; b.S = S; b.d_pts = d_pts; b.d_dirs = d_dirs;
    for (int i = 0; i < 2; ++i) b.grads.sigma_w[i] = grads->sigma_w[i];
    for (int i = 0; i < 3; ++i) { b.grads.color_w[i] = grads->color_w[i]; b.grads.color_b[i] = grads->color_b[i]; }
    b.accumulate = grads->accumulate ? 1 : 0;
    return run_voxel_backward_dispatch(precision, v->hidden_dim, b, as_stream(stream));
}

// ---- training: the grids as parameters -------------------------------------------------------------------------------
int evd_voxel_grid_sizes(const evd_voxel* v, long* sizes) {
    EVD_REQUIRE(v && sizes, "evd_voxel_grid_sizes: null argument");
    for (int i = 0; i < 3; ++i) {
        sizes[i] = (long)(v->plane[i].bytes / sizeof(float));
        sizes[3 + i] = (long)(v->line[i].bytes / sizeof(float));
    }
    sizes[6] = (long)(v->basis.bytes / sizeof(float));
    return EVD_OK;
}

int evd_voxel_get_grids(const evd_voxel* v, float* const* plane, float* const* line, float* basis, void* stream) {
    EVD_REQUIRE(v && plane && line && basis, "evd_voxel_get_grids: null argument");
    hipStream_t st = as_stream(stream);
    for (int i = 0; i < 3; ++i) {
        EVD_HIP(hipMemcpyAsync(plane[i], v->plane[i].p, v->plane[i].bytes, hipMemcpyDeviceToDevice, st));
        EVD_HIP(hipMemcpyAsync(line[i], v->line[i].p, v->line[i].bytes, hipMemcpyDeviceToDevice, st));
    }
    EVD_HIP(hipMemcpyAsync(basis, v->basis.p, v->basis.bytes, hipMemcpyDeviceToDevice, st));
    return EVD_OK;
}

// the seven tensors of a level in ONE launch: every source value is read once and written twice, as the float32 copy the float32-grade modes
// and the backward gather, and as the float16 copy the half-precision modes gather (13 launches and a second read of 165 MB before)
struct GridLoadSeg { const float* src; float* dst; _Float16* dst_h; long n4, tail0, n; };
struct GridLoadSegs { GridLoadSeg s[7]; };
static __global__ __launch_bounds__(256) void k_load_grids(const GridLoadSegs segs) {
    const GridLoadSeg s = segs.s[blockIdx.y];
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < s.n4; i += stride) {
        const f32x4 v = reinterpret_cast<const f32x4*>(s.src)[i];
        reinterpret_cast<f32x4*>(s.dst)[i] = v;
        if (s.dst_h) {
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            const h4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            reinterpret_cast<h4*>(s.dst_h)[i] = h;
        }
    }
    for (long i = s.tail0 + (long)blockIdx.x * 256 + threadIdx.x; i < s.n; i += stride) {
        s.dst[i] = s.src[i];
        if (s.dst_h) s.dst_h[i] = (_Float16)s.src[i];
    }
}

int evd_voxel_load_grids(evd_voxel* v, const float* const* plane, const float* const* line, const float* basis, void* stream) {
    EVD_REQUIRE(v && plane && line && basis, "evd_voxel_load_grids: null argument");
    hipStream_t st = as_stream(stream);
    GridLoadSegs segs;
    long nmax = 0;
    auto seg = [&](int k, const float* src, DevBuf& dst, DevBuf* dst_h) {
        const long n = (long)(dst.bytes / sizeof(float));
        const bool vec = ((uintptr_t)src % 16) == 0;            // device allocations are 256-byte aligned; a caller's slice may not be
        segs.s[k] = GridLoadSeg{src, (float*)dst.p, dst_h ? (_Float16*)dst_h->p : nullptr, vec ? n / 4 : 0, vec ? (n / 4) * 4 : 0, n};
        nmax = n > nmax ? n : nmax;
    };
    for (int i = 0; i < 3; ++i) {
        EVD_REQUIRE(plane[i] && line[i], "evd_voxel_load_grids: missing grid %d", i);
        seg(i, plane[i], v->plane[i], &v->plane_h[i]);
        seg(3 + i, line[i], v->line[i], &v->line_h[i]);
    }
    seg(6, basis, v->basis, nullptr);
    const long bx = cdiv(nmax / 4 + 1, 256L) < 2048 ? cdiv(nmax / 4 + 1, 256L) : 2048;
    hipLaunchKernelGGL(k_load_grids, dim3((unsigned)bx, 7), dim3(256), 0, st, segs);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

int evd_voxel_sample_bwd(const evd_voxel* v, const float* pts, long n, const float* d_out, int d_stride, int d_col,
                         const evd_voxel_grid_grads* g, float* d_pts, void* stream) {
    EVD_REQUIRE(v && pts && d_out && g && n >= 0 && d_stride >= d_col + v->app_dim, "evd_voxel_sample_bwd: bad arguments");
    EVD_REQUIRE(v->app_act == EVD_ACT_NONE, "evd_voxel_sample_bwd: only app_actfn none is built (all shipped configs)");
    if (n == 0) return EVD_OK;
    GridGrads gg;
    for (int i = 0; i < 3; ++i) { gg.plane[i] = g->plane[i]; gg.line[i] = g->line[i]; }
    gg.basis = g->basis;
    return launch_voxel_sample_bwd(v->gp, pts, n, d_out, d_stride, d_col, gg, d_pts, as_stream(stream));
}

// developer switch: EVD_SCATTER=binned selects the sort + LDS-tile form behind evd_voxel_sample_bwd_ws (default: the hybrid form)
static bool scatter_binned() { static const bool b = [] { const char* e = getenv("EVD_SCATTER"); return e && !strcmp(e, "binned"); }(); return b; }
size_t evd_voxel_sample_bwd_workspace_bytes(const evd_voxel* v, long n) {
    if (!v || n <= 0) return 0;
    return scatter_binned() ? voxel_scatter_workspace_bytes(v->gp, n) : voxel_scatter_hybrid_workspace_bytes(v->gp, n);
}

static int sample_bwd_ws(const evd_voxel* v, bool half_grids, const float* pts, long n, const float* d_out, int d_stride, int d_col,
                         const evd_voxel_grid_grads* g, float* d_pts, void* workspace, size_t workspace_bytes, void* stream);
int evd_voxel_sample_bwd_ws(const evd_voxel* v, const float* pts, long n, const float* d_out, int d_stride, int d_col,
                            const evd_voxel_grid_grads* g, float* d_pts, void* workspace, size_t workspace_bytes, void* stream) {
    return sample_bwd_ws(v, false, pts, n, d_out, d_stride, d_col, g, d_pts, workspace, workspace_bytes, stream);
}
int evd_voxel_sample_bwd_prec(const evd_voxel* v, int precision, const float* pts, long n, const float* d_out, int d_stride, int d_col,
                              const evd_voxel_grid_grads* g, float* d_pts, void* workspace, size_t workspace_bytes, void* stream) {
    EVD_REQUIRE(precision >= 0 && precision <= EVD_PREC_F16M, "evd_voxel_sample_bwd_prec: unknown precision %d", precision);
    return sample_bwd_ws(v, v && grids_half_for(v, precision), pts, n, d_out, d_stride, d_col, g, d_pts, workspace, workspace_bytes, stream);
}
static int sample_bwd_ws(const evd_voxel* v, bool half_grids, const float* pts, long n, const float* d_out, int d_stride, int d_col,
                         const evd_voxel_grid_grads* g, float* d_pts, void* workspace, size_t workspace_bytes, void* stream) {
    EVD_REQUIRE(v && pts && d_out && g && n >= 0 && d_stride >= d_col + v->app_dim, "evd_voxel_sample_bwd_ws: bad arguments");
    EVD_REQUIRE(v->app_act == EVD_ACT_NONE, "evd_voxel_sample_bwd_ws: only app_actfn none is built (all shipped configs)");
    if (n == 0) return EVD_OK;
    GridGrads gg;
    for (int i = 0; i < 3; ++i) { gg.plane[i] = g->plane[i]; gg.line[i] = g->line[i]; }
    gg.basis = g->basis;
    if (workspace && workspace_bytes > 0 && voxel_scatter_binned_ok(v->gp, gg, n)) {
        if (scatter_binned()) return launch_voxel_sample_bwd_binned(v->gp, pts, n, d_out, d_stride, d_col, gg, d_pts, workspace, workspace_bytes, as_stream(stream));
        return launch_voxel_sample_bwd_hybrid(v->gp, pts, n, d_out, d_stride, d_col, gg, d_pts, workspace, workspace_bytes, as_stream(stream), half_grids);
    }
    return launch_voxel_sample_bwd(v->gp, pts, n, d_out, d_stride, d_col, gg, d_pts, as_stream(stream));
}

int evd_voxel_tv_loss_bwd(const evd_voxel* v, const float* d_loss, const evd_voxel_grid_grads* g, void* stream) {
    EVD_REQUIRE(v && g && d_loss, "evd_voxel_tv_loss_bwd: null argument");
    hipStream_t st = as_stream(stream);
    TvJobs jobs;
    jobs.n = 6;
    for (int i = 0; i < 3; ++i) {
        const int C = v->n_comp[i], Wp = v->grid[kMat0[i]], Hp = v->grid[kMat1[i]], Lp = v->grid[kVec[i]];
        jobs.j[i] = TvJob{(const float*)v->plane[i].p, g->plane[i], Hp, Wp, C, 1e-2f, 0, 0};
        jobs.j[3 + i] = TvJob{(const float*)v->line[i].p, g->line[i], Lp, 1, C, 1e-3f, 0, 0};
    }
    return launch_tv_bwd_level(jobs, d_loss, st);
}

int evd_voxel_tv_loss(const evd_voxel* v, float* out, void* stream) {
    EVD_REQUIRE(v && out, "evd_voxel_tv_loss: null argument");
    hipStream_t st = as_stream(stream);
    double* acc = (double*)v->tv_acc.p;
    TvShape s;
    TvJobs jobs;
    jobs.n = 6;
    for (int i = 0; i < 3; ++i) {
        const int C = v->n_comp[i], Wp = v->grid[kMat0[i]], Hp = v->grid[kMat1[i]], Lp = v->grid[kVec[i]];
        jobs.j[i] = TvJob{(const float*)v->plane[i].p, nullptr, Hp, Wp, C, 1e-2f, 0, 0};
        jobs.j[3 + i] = TvJob{(const float*)v->line[i].p, nullptr, Lp, 1, C, 1e-3f, 0, 0};
    }
    int rc = launch_tv_level(jobs, acc, &s, st);
    if (rc) return rc;
    return launch_tv_finish(acc, s, out, st);
}

}  // extern "C"
