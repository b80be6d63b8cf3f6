#include <atomic>
// C-ABI entry points of the NeRF backbone: weight packing, fused MLP launch, render_rays orchestration.
#include "evd_common.h"
#include "nerf_mlp.h"
#include "nerf_mlp_kernel.h"
#include "nerf_net.h"
#include "nerf_train.h"
#include "pack.h"

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace evd {

char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

int nerf_mlp_generic_dispatch(int prec, int W, const MlpParams& p, hipStream_t st);
int nerf_mlp_pipe_dispatch(int prec, const MlpParams& p, hipStream_t st);
int nerf_mlp_c_dispatch(int W, int D, int skip, const MlpParams& p, hipStream_t st);       // kernel_nerf_mlp.hip: compensated float16 mode
int nerf_mlp_c_chunks(int W, int D, int skip);                                             // 0: not built for this network

// the side-stream test hook (evd_common.h): ONE copy of the spin kernel in the library, the launch checked, launches counted
static std::atomic<long> g_side_spins{0};
static __global__ void k_test_spin(long long ticks) {          // wall_clock64: the constant 100 MHz counter
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
int test_side_spin(hipStream_t side) {
    static const int us = [] { const char* e = getenv("EVD_TEST_SIDE_SPIN_US"); return e ? atoi(e) : 0; }();
    if (us <= 0) return EVD_OK;
    hipLaunchKernelGGL(k_test_spin, dim3(1), dim3(64), 0, side, (long long)us * 100);
    EVD_LAUNCH_CHECK();
    g_side_spins.fetch_add(1, std::memory_order_relaxed);
    return EVD_OK;
}
bool test_skip_side_join() {
    static const bool skip = [] { const char* e = getenv("EVD_TEST_SKIP_SIDE_JOIN"); return e && e[0] == '1'; }();
    return skip;
}

}  // namespace evd

using namespace evd;

extern "C" {

const char* evd_last_error(void) { return evd::err_buf(); }
long evd_debug_side_spin_count(void) { return evd::g_side_spins.load(std::memory_order_relaxed); }
int evd_version(void) { return 120; }      // 110: training entries (evd_*_train, evd_*_backward, evd_*_load_params, ...); 120: evd_awp_embed_*, the f16x3 training mode (evd_*_train_store_bytes_prec), evd_voxel_mlp_backward(awp_store), evd_voxel_sample_bwd_ws, evd_merge_features

int evd_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return fail(EVD_E_NODEVICE, "hipGetDeviceCount failed");
    return n;
}

// Every stream keeps, next to its bytes, the index map element -> parameter arena (pack.h), so that new parameter values
// are re-packed on the device (evd_nerf_load_params) with the arithmetic the host packer used at creation.
static int upload_stream(evd_nerf::Packed& dst, const StreamBuilder& sb) { return dst.upload(sb); }

// canonical parameter order of the arena: pts_linears[l].{weight, bias} for l < D, then views_linears.0, feature_linear,
// alpha_linear, rgb_linear ({weight, bias} each)
static void nerf_param_sizes(int D, int W, int skip, int L, int Lv, long* sz) {
    const int IC = 3 * (1 + 2 * L), ICV = 3 * (1 + 2 * Lv);
    for (int l = 0; l < D; ++l) {
        sz[2 * l] = (long)W * (l == 0 ? IC : (l - 1 == skip ? W + IC : W));
        sz[2 * l + 1] = W;
    }
    long* h = sz + 2 * D;
    h[0] = (long)(W / 2) * (W + ICV); h[1] = W / 2;      // views
    h[2] = (long)W * W; h[3] = W;                        // feature
    h[4] = W; h[5] = 1;                                  // alpha
    h[6] = 3L * (W / 2); h[7] = 3;                       // rgb
}

int evd_nerf_create(const evd_nerf_desc* d, evd_nerf** out) {
    EVD_REQUIRE(d && out, "evd_nerf_create: null argument");
    // frequency counts other than (PE_L, PE_LV) run in the generic kernel only (no pipelined / f16c / training streams)
    EVD_REQUIRE(d->multires >= 0 && d->multires <= PE_L_MAX && d->multires_views >= 0 && pe_ksteps(d->multires_views) <= 4,
                "evd_nerf_create: multires %d / multires_views %d out of range (0..%d / 0..10)", d->multires, d->multires_views, PE_L_MAX);
    const int L = d->multires, Lv = d->multires_views, VKS = pe_ksteps(Lv) <= 2 ? 2 : 4;
    const bool standard = L == PE_L && Lv == PE_LV;
    EVD_REQUIRE(d->W == 256 || d->W == 64, "evd_nerf_create: netwidth %d not built (64, 256)", d->W);
    EVD_REQUIRE(d->D >= 1 && d->D <= EVD_MAX_LAYERS, "evd_nerf_create: netdepth %d out of range", d->D);
    const bool no_views = !d->views_w && !d->feature_w && !d->alpha_w && !d->rgb_w && d->output_w;
    EVD_REQUIRE(no_views || (d->views_w && d->feature_w && d->alpha_w && d->rgb_w),
                "evd_nerf_create: give either the view branch (views / feature / alpha / rgb) or output_linear (use_viewdirs=False)");
    EVD_REQUIRE(!no_views || (d->output_b && (d->output_ch == 4 || d->output_ch == 5)), "evd_nerf_create: output_linear needs its bias and output_ch 4 or 5");
    const int W = d->W, T = W / 32, KS = W / 16, IC = 3 * (1 + 2 * L), ICV = 3 * (1 + 2 * Lv), D = d->D;
    evd_nerf* n = new evd_nerf();
    n->multires = L; n->multires_views = Lv;
    n->D = D; n->W = W; n->skip = d->skip; n->rgb_act = d->rgb_act; n->sigma_act = d->sigma_act; n->rmnear = d->rmnear;
    n->no_views = no_views ? 1 : 0;

    // host copy of every parameter in one arena (missing biases = zeros); the packers below read from it
    long sz[2 * EVD_MAX_LAYERS + 8];
    nerf_param_sizes(D, W, d->skip, L, Lv, sz);
    n->nparam_blocks = 2 * D + 8;
    if (no_views) {                     // canonical order: pts_linears[l].{weight, bias}, then output_linear.{weight, bias}
        sz[2 * D] = (long)d->output_ch * W; sz[2 * D + 1] = d->output_ch;
        n->nparam_blocks = 2 * D + 2;
    }
    long total = 0;
    for (int i = 0; i < n->nparam_blocks; ++i) { n->param_off[i] = total; total += sz[i]; }
    n->param_off[n->nparam_blocks] = total;
    std::vector<float> arena((size_t)total, 0.f);
    {
        const float* srcs[2 * EVD_MAX_LAYERS + 8];
        for (int l = 0; l < D; ++l) { srcs[2 * l] = d->pts_w[l]; srcs[2 * l + 1] = d->pts_b[l]; }
        const float* heads[8] = {d->views_w, d->views_b, d->feature_w, d->feature_b, d->alpha_w, d->alpha_b, d->rgb_w, d->rgb_b};
        for (int i = 0; i < 8; ++i) srcs[2 * D + i] = heads[i];
        if (no_views) { srcs[2 * D] = d->output_w; srcs[2 * D + 1] = d->output_b; }
        for (int i = 0; i < n->nparam_blocks; ++i)
            if (srcs[i]) memcpy(arena.data() + n->param_off[i], srcs[i], sz[i] * sizeof(float));
    }
    const float* A = arena.data();
    auto P = [&](int i) { return A + n->param_off[i]; };
    auto pts_w = [&](int l) { return P(2 * l); };
    const float *views_w = P(2 * D), *feature_w = no_views ? nullptr : P(2 * D + 2), *alpha_w = no_views ? nullptr : P(2 * D + 4), *rgb_w = no_views ? nullptr : P(2 * D + 6);
    const float* output_w = P(2 * D);
    const int out_rows = d->output_ch < 4 ? d->output_ch : 4;       // raw2outputs reads channels 0..3

    auto pe_col = [L](int j, int kk) { return pe_src_col(L, 8 * j + (kk & 7), kk >> 3); };
    auto hid_col = [](int j, int kk) { return 16 * j + phi(kk); };
    auto views_col = [&](int j, int kk) {
        if (j < KS) return hid_col(j, kk);
        const int c = pe_src_col(Lv, 8 * (j - KS) + (kk & 7), kk >> 3);
        return c < 0 ? -1 : W + c;
    };
    // pdh < 0: the skip layer's k-steps are [pe_0..3 | h_0..h_{KS-1}] (generic kernel);
    // else [h_0..h_{pdh-1} | pe_0..3 | h_pdh..h_{KS-1}] (pipelined kernel, nerf_mlp_kernel.h)
    auto build = [&](StreamBuilder& sb, int pdh) {
        sb.layer(pts_w(0), W, IC, T, PE_KS, true, pe_col);
        for (int l = 1; l < D; ++l) {
            if (l - 1 == d->skip) {
                auto wide_col = [&](int j, int kk) {
                    if (pdh < 0) return j < PE_KS ? pe_col(j, kk) : IC + hid_col(j - PE_KS, kk);
                    if (j < pdh) return IC + hid_col(j, kk);
                    if (j < pdh + PE_KS) return pe_col(j - pdh, kk);
                    return IC + hid_col(j - PE_KS, kk);
                };
                sb.layer(pts_w(l), W, W + IC, T, PE_KS + KS, true, wide_col);
            } else {
                sb.layer(pts_w(l), W, W, T, KS, true, hid_col);
            }
        }
        if (no_views) {
            sb.layer(output_w, out_rows, W, 1, KS, true, hid_col);
            return;
        }
        sb.layer(alpha_w, 1, W, 1, KS, false, hid_col);
        sb.layer(feature_w, W, W, T, KS, false, hid_col);
        sb.layer(views_w, W / 2, W + ICV, T / 2, KS + VKS, false, views_col);
        sb.layer(rgb_w, 3, W / 2, 1, KS / 2, true, hid_col);
    };
    int rc = EVD_OK;
    for (int prec = 0; prec < EVD_NUM_PREC && !rc; ++prec) {
        if (prec == EVD_PREC_F16C) {      // compensated float16 mode: its own stream (float16 + fp6 fragments) and row scales; pipelined kernel only
            n->nchunks[prec] = n->pipe_chunks[prec] = 0;
            if (no_views || !standard || !nerf_mlp_c_chunks(W, D, d->skip)) continue;
            StreamBuilderC sc(PIPE_CB);
            sc.arena = A;
            const int KB = KS / 4;
            sc.layer(pts_w(0), W, IC, T, PE_KS, 2, pe_col);
            for (int l = 1; l < D; ++l) {
                if (l - 1 == d->skip) {       // blocks [h_0 .. h_{KB-2} | pe | h_{KB-1}]
                    auto wide_col = [&](int j, int kk) {
                        if (j < 4 * (KB - 1)) return IC + c_hid_col(j, kk);
                        if (j < 4 * (KB - 1) + PE_KS) return pe_col(j - 4 * (KB - 1), kk);
                        return IC + c_hid_col(j - PE_KS, kk);
                    };
                    sc.layer(pts_w(l), W, W + IC, T, PE_KS + KS, 2, wide_col);
                } else {
                    sc.layer(pts_w(l), W, W, T, KS, 2, c_hid_col);
                }
            }
            sc.layer(alpha_w, 1, W, 1, KS, 1, c_hid_col);
            sc.layer(feature_w, W, W, T, KS, 2, c_hid_col);
            sc.layer(views_w, W / 2, W + ICV, T / 2, KS + PEV_KS, 2, [&](int j, int kk) {
                if (j < KS) return c_hid_col(j, kk);
                const int c = pe_src_col(Lv, 8 * (j - KS) + (kk & 7), kk >> 3);
                return c < 0 ? -1 : W + c;
            });
            sc.layer(rgb_w, 3, W / 2, 1, KS / 2, 1, c_hid_col);
            n->pipe_chunks[prec] = (int)(sc.bytes.size() / PIPE_CB);
            if (n->pipe_chunks[prec] != nerf_mlp_c_chunks(W, D, d->skip)) rc = fail(EVD_E_INVALID, "evd_nerf_create: f16c stream has %d chunks, kernel expects %d", n->pipe_chunks[prec], nerf_mlp_c_chunks(W, D, d->skip));
            if (!rc) rc = n->pipe_c.upload(sc);
            continue;
        }
        StreamBuilder sb(prec);
        sb.arena = A;
        build(sb, -1);
        n->nchunks[prec] = (int)(sb.bytes.size() / chunk_bytes(prec));
        rc = upload_stream(n->stream[prec], sb);
        n->pipe_chunks[prec] = 0;
        if (!rc && !no_views && standard && nerf_pipe_built(prec, W, D, d->skip)) {
            StreamBuilder sp(prec, PIPE_CB);
            sp.arena = A;
            sp.group = nerf_group(prec);
            build(sp, KS - 2 * nerf_group(prec));
            n->pipe_chunks[prec] = (int)(sp.bytes.size() / PIPE_CB);
            rc = upload_stream(n->pipe[prec], sp);
        }
        // training path (bf16 / f16 / split-f16 on the pipelined network): W^T streams of the dgrad chain, one per layer (nerf_train_kernel.h)
        if (rc || !is_train_prec(prec) || !n->pipe_chunks[prec]) continue;
        auto put = [&](int which, auto fill) {
            StreamBuilder sb2(prec, PIPE_CB);
            sb2.arena = A;
            sb2.group = 1;
            fill(sb2);
            return upload_stream(n->bwd[prec][which], sb2);
        };
        rc = put(EVD_BWD_RGB, [&](StreamBuilder& b) { b.layer_transposed(rgb_w, 3, W / 2, 0, W / 2, nullptr, 0, T / 2, 1, true, [](int, int kk) { return kk < 3 ? kk : -1; }); });
        if (!rc) rc = put(EVD_BWD_VIEWS, [&](StreamBuilder& b) { b.layer_transposed(views_w, W / 2, W + ICV, 0, W, nullptr, 0, T, KS / 2, true, hid_col); });
        if (!rc) rc = put(EVD_BWD_HEAD, [&](StreamBuilder& b) {
            b.layer_transposed(feature_w, W, W, 0, W, alpha_w, 1, T, KS + 1, true, [&](int j, int kk) { return j < KS ? hid_col(j, kk) : (kk == 0 ? W : -1); });
        });
        // encoding rows (for the gradient w.r.t. the rays): output row idx <-> encoding column so that the output fragments come out in
        // the encoding's own arrangement (fragment j, position kk <-> pe_src_col(L, 8 j + (kk & 7), kk >> 3))
        auto enc_row = [](int L_, int idx) { const int j = idx / 16, kk = phi_inv(idx % 16); return pe_src_col(L_, 8 * j + (kk & 7), kk >> 3); };
        auto enc_layer = [&](StreamBuilder& b, const float* Wm, int in_dim, int col0, int L_, int tiles, int ksteps) {
            b.layer_at(tiles, ksteps, true, [&](int t, int r) { const int c = enc_row(L_, 32 * t + r); return c < 0 ? -1 : col0 + c; }, hid_col,
                       [=](int r, int c) { return Wm + (size_t)c * in_dim + r; });
        };
        if (!rc) rc = put(EVD_BWD_PE0, [&](StreamBuilder& b) { enc_layer(b, pts_w(0), IC, 0, PE_L, 2, KS); });
        if (!rc && d->skip >= 0 && d->skip + 1 < D) rc = put(EVD_BWD_PESKIP, [&](StreamBuilder& b) { enc_layer(b, pts_w(d->skip + 1), W + IC, 0, PE_L, 2, KS); });
        if (!rc) rc = put(EVD_BWD_DIR, [&](StreamBuilder& b) { enc_layer(b, views_w, W + ICV, W, PE_LV, 1, KS / 2); });
        for (int l = 1; l < D && !rc; ++l) {
            const bool wide = l - 1 == d->skip;
            rc = put(EVD_BWD_HIDDEN1 + l - 1, [&](StreamBuilder& b) { b.layer_transposed(pts_w(l), W, wide ? W + IC : W, wide ? IC : 0, W, nullptr, 0, T, KS, true, hid_col); });
        }
    }
    if (rc) { evd_nerf_destroy(n); return rc; }
    {   // wgrad index maps (nerf_train.h): fragment column (fragment j = i / 16, position kk = i % 16) -> parameter row / column
        std::vector<int> m(MAP_TOTAL, -1);
        for (int i = 0; i < 256; ++i) {
            m[MAP_HID + i] = hid_col(i / 16, i % 16);
            m[MAP_HID_SKIP + i] = IC + hid_col(i / 16, i % 16);
        }
        for (int i = 0; i < 64; ++i) m[MAP_PE + i] = pe_col(i / 16, i % 16);
        for (int i = 0; i < 32; ++i) {
            const int c = pe_src_col(Lv, 8 * (i / 16) + (i % 16 & 7), (i % 16) >> 3);
            m[MAP_DIR + i] = c < 0 ? -1 : W + c;
        }
        for (int i = 0; i < 3; ++i) m[MAP_RGB + i] = i;
        m[MAP_ALPHA] = 0;
        if ((rc = n->wmaps.upload(m.data(), m.size() * sizeof(int)))) { evd_nerf_destroy(n); return rc; }
    }
    // biases, one 32-float row block per output tile, in stream order
    std::vector<float> b;
    std::vector<int32_t> bsrc;
    auto push = [&](int block, int out_dim, int tiles) {
        for (int i = 0; i < tiles * 32; ++i) {
            b.push_back(i < out_dim ? P(block)[i] : 0.f);
            bsrc.push_back(i < out_dim ? (int32_t)(n->param_off[block] + i) : -1);
        }
    };
    for (int l = 0; l < D; ++l) push(2 * l + 1, W, T);
    if (no_views) {
        push(2 * D + 1, out_rows, 1);
    } else {
        push(2 * D + 5, 1, 1);
        push(2 * D + 3, W, T);
        push(2 * D + 1, W / 2, T / 2);
        push(2 * D + 7, 3, 1);
    }
    rc = n->bias.upload(b.data(), b.size() * sizeof(float));
    if (!rc) rc = n->bias_src.upload(bsrc.data(), bsrc.size() * sizeof(int32_t));
    if (rc) { evd_nerf_destroy(n); return rc; }
    *out = n;
    return EVD_OK;
}

void evd_nerf_destroy(evd_nerf* n) {
    if (!n) return;
    for (int i = 0; i < EVD_NUM_PREC; ++i) {
        n->stream[i].release();
        n->pipe[i].release();
        for (int k = 0; k < EVD_BWD_NSTREAMS; ++k) n->bwd[i][k].release();
    }
    n->wmaps.release();
    n->batch.release();
    n->bias.release();
    n->bias_src.release();
    n->pipe_c.release();
    n->side.release();
    delete n;
}

long evd_nerf_param_count(const evd_nerf* net) { return net ? net->param_off[net->nparam_blocks] : 0; }

int evd_nerf_param_blocks(const evd_nerf* net, long* offsets, int capacity) {
    EVD_REQUIRE(net, "evd_nerf_param_blocks: null network");
    if (offsets)
        for (int i = 0; i <= net->nparam_blocks && i < capacity; ++i) offsets[i] = net->param_off[i];
    return net->nparam_blocks;
}

int evd_nerf_load_params(evd_nerf* net, const float* params, void* stream) {
    EVD_REQUIRE(net && params, "evd_nerf_load_params: null argument");
    hipStream_t st = as_stream(stream);
    int rc;
    std::vector<PackedStream*> all;
    for (int i = 0; i < EVD_NUM_PREC; ++i) {
        all.push_back(&net->stream[i]);
        all.push_back(&net->pipe[i]);
        for (int k = 0; k < EVD_BWD_NSTREAMS; ++k) all.push_back(&net->bwd[i][k]);
    }
    if ((rc = repack_batch(net->batch, all, params, st))) return rc;
    if ((rc = repack_stream_c(net->pipe_c, params, st))) return rc;
    const long nb = (long)(net->bias.bytes / sizeof(float));
    hipLaunchKernelGGL(k_gather_f32, dim3((unsigned)cdiv(nb, 256L)), dim3(256), 0, st, params, (const int*)net->bias_src.p, nb, (float*)net->bias.p);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

size_t evd_nerf_stream_bytes(const evd_nerf* net, int precision) {
    if (!net || precision < 0 || precision >= EVD_NUM_PREC) return 0;
    if (precision == EVD_PREC_F16C) return net->pipe_c.data.bytes;
    return net->pipe_chunks[precision] ? net->pipe[precision].data.bytes : net->stream[precision].data.bytes;
}

int evd_nerf_mlp(const evd_nerf* net, int precision, const float* ray_batch, const float* z, long R, int S,
                 float* raw, float* feature, int feature_kind, void* stream) {
    EVD_REQUIRE(net && ray_batch && z && raw, "evd_nerf_mlp: null argument");
    EVD_REQUIRE(precision >= 0 && precision < EVD_NUM_PREC, "evd_nerf_mlp: unknown precision %d", precision);
    EVD_REQUIRE(R >= 0 && S >= 1, "evd_nerf_mlp: bad shape R=%ld S=%d", R, S);
    EVD_REQUIRE(!feature || feature_kind == 1 || feature_kind == 2, "evd_nerf_mlp: feature_kind must be 1 or 2");
    if (R == 0) return EVD_OK;
    MlpParams p{};
    static const bool no_pipe = env_flag("EVD_NO_PIPE");      // developer switch: force the generic kernel
    const bool piped = net->pipe_chunks[precision] > 0 && !no_pipe;
    p.wstream = (const char*)(piped ? net->pipe[precision].data.p : net->stream[precision].data.p);
    p.bias = (const float*)net->bias.p;
    p.ray_batch = ray_batch; p.z = z; p.nsamp = R * (long)S; p.S = S; p.ncol = net->no_views ? 8 : 11; p.no_views = net->no_views;
    p.pe_l = net->multires; p.pe_lv = net->multires_views;
    EVD_REQUIRE(!(net->no_views && feature && feature_kind == 1), "evd_nerf_mlp: a use_viewdirs=False network has no after_linear feature (nerf.py:159)");
    EVD_REQUIRE(!net->no_views || precision != EVD_PREC_F16C, "evd_nerf_mlp: EVD_PREC_F16C is not built for use_viewdirs=False networks");
    p.D = net->D; p.skip = net->skip; p.nchunks = piped ? net->pipe_chunks[precision] : net->nchunks[precision]; p.nbias = (int)(net->bias.bytes / sizeof(float));
    p.raw = raw; p.feature = feature; p.feature_kind = feature ? feature_kind : 0; p.act = nullptr; p.wscale = nullptr;
    if (precision == EVD_PREC_F16C) {
        EVD_REQUIRE(net->pipe_chunks[precision] > 0, "evd_nerf_mlp: EVD_PREC_F16C is built for netdepth 8, netwidth 256, skips [4], multires 10 / 4 only");
        EVD_REQUIRE(!feature, "evd_nerf_mlp: EVD_PREC_F16C has no feature-row variant (use EVD_PREC_F16X3)");
        p.wstream = (const char*)net->pipe_c.data.p;
        p.nchunks = net->pipe_chunks[precision];
        p.wscale = (const unsigned*)net->pipe_c.scales.p;
        return nerf_mlp_c_dispatch(net->W, net->D, net->skip, p, as_stream(stream));
    }
    if (piped) return nerf_mlp_pipe_dispatch(precision, p, as_stream(stream));
    return nerf_mlp_generic_dispatch(precision, net->W, p, as_stream(stream));
}

// ------------------------------------------------------------------------------------------------
// NeRFAll.render + render_rays, mode='nerf' (networks/renderer.py:399-466, 129-264 else-branch)
static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

size_t evd_nerf_render_workspace_bytes(const evd_render_cfg* cfg, long R) {
    if (!cfg || R < 0) return 0;
    const size_t S = cfg->N_samples, Ni = cfg->N_importance > 0 ? cfg->N_importance : 0, St = S + Ni, r = (size_t)R;
    size_t b = 0;
    b += align256(r * 11 * 4);          // ray_batch
    b += align256(r * St * 4);          // z (merged)
    b += align256(r * S * 4);           // z0
    b += align256(r * St * 16);         // raw
    b += align256(r * St * 4);          // weights
    b += align256(r * S * 4);           // weights0
    b += align256(r * (Ni ? Ni : 1) * 4);  // z_samples
    return b + 256;
}

}  // extern "C"
int evd_ray_batch_z(const evd_render_cfg* cfg, const float* rays, long R, const float* t_rand, float* ray_batch, float* z, void* stream);   // kernels_render.hip
// where the first pass's z lives (an output the caller asked for, else the workspace slot): shared by the two entries below
static float* render_z_slot(const evd_render_cfg* cfg, long R, const evd_render_out* out, void* workspace) {
    const int S = cfg->N_samples, Ni = cfg->N_importance > 0 ? cfg->N_importance : 0, St = S + Ni;
    char* w = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    w += align256((size_t)R * 11 * 4) + align256((size_t)R * St * 4);
    float* z0 = (float*)w;
    return Ni ? (out->z_vals0 ? out->z_vals0 : z0) : (out->z_vals ? out->z_vals : z0);
}
static int render_rays_impl(const evd_nerf* coarse, const evd_nerf* fine, const evd_render_cfg* cfg, const float* rb,
                            long R, const float* t_rand, const float* u, const float* noise0, const float* noise1,
                            evd_render_out* out, void* workspace, size_t workspace_bytes, void* stream, bool z_done) {
    EVD_REQUIRE(coarse && cfg && out && (rb || R == 0), "evd_nerf_render_rays: null argument");
    EVD_REQUIRE((cfg->use_viewdirs != 0) == (coarse->no_views == 0) && (!fine || fine->no_views == coarse->no_views),
                "evd_nerf_render_rays: cfg->use_viewdirs does not match the networks (view branch vs output_linear)");
    const int nc = cfg->use_viewdirs ? 11 : 8;
    EVD_REQUIRE(cfg->N_samples >= 1, "evd_nerf_render_rays: N_samples must be >= 1");
    EVD_REQUIRE(cfg->N_importance <= 0 || fine, "evd_nerf_render_rays: N_importance > 0 needs the fine network");
    EVD_REQUIRE(cfg->N_importance <= 0 || cfg->N_samples >= 3, "evd_nerf_render_rays: hierarchical sampling needs N_samples >= 3");
    EVD_REQUIRE(!(cfg->perturb > 0.f) || (t_rand && (cfg->N_importance <= 0 || u)),
                "evd_nerf_render_rays: perturb > 0 needs explicit t_rand (and u) draws");
    if (R == 0) return EVD_OK;
    const size_t need = evd_nerf_render_workspace_bytes(cfg, R);
    if (!workspace || workspace_bytes < need)
        return fail(EVD_E_WORKSPACE, "evd_nerf_render_rays: workspace %zu < %zu bytes", workspace_bytes, need);
    const int S = cfg->N_samples, Ni = cfg->N_importance > 0 ? cfg->N_importance : 0, St = S + Ni;
    char* w = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    auto take = [&](size_t bytes) { char* p = w; w += align256(bytes); return (float*)p; };
    const size_t r = (size_t)R;
    (void)take(r * 11 * 4);             // ray_batch slot (used by evd_nerf_render)
    float* z2 = take(r * St * 4);
    float* z0 = take(r * S * 4);
    float* raw = take(r * St * 16);
    float* wts = take(r * St * 4);
    float* wts0 = take(r * S * 4);
    float* zs = take(r * (Ni ? Ni : 1) * 4);
    int rc;
    float* zc = (Ni ? (out->z_vals0 ? out->z_vals0 : z0) : (out->z_vals ? out->z_vals : z0));
    // the whole step in ONE launch where the compensated float16 kernel can take it: z stratification in its prologue, raw2outputs in
    // its epilogue (nerf_mlp_c_kernel.h FUSE); z / raw / weights reach HBM only where the caller asked for them
    // Measured (4096 x 128): 0.869 ms per step against 0.864 with the separate kernels -- at one wavefront per SIMD the serial prologue /
    // epilogue costs more than the two small launches it saves, so it is opt-in (EVD_FUSE_STEP=1); tests/test_gpu_fullsize.py covers it.
    const bool no_fuse = !env_flag("EVD_FUSE_STEP") || z_done;
    if (!Ni && cfg->precision == EVD_PREC_F16C && coarse->pipe_chunks[EVD_PREC_F16C] > 0 && (S == 32 || S == 64 || S == 128) && !noise0 &&
        !out->feature && !no_fuse && !(!cfg->is_train && coarse->rmnear > 0.f)) {
        MlpParams p{};
        p.wstream = (const char*)coarse->pipe_c.data.p;
        p.nchunks = coarse->pipe_chunks[EVD_PREC_F16C];
        p.wscale = (const unsigned*)coarse->pipe_c.scales.p;
        p.bias = (const float*)coarse->bias.p;
        p.nbias = (int)(coarse->bias.bytes / sizeof(float));
        p.ray_batch = rb; p.z = nullptr; p.nsamp = R * (long)S; p.S = S; p.ncol = 11; p.D = coarse->D; p.skip = coarse->skip;
        p.raw = out->raw; p.feature = nullptr; p.feature_kind = 0; p.act = nullptr;
        p.fuse = 1; p.lindisp = cfg->lindisp; p.perturb = cfg->perturb > 0.f; p.t_rand = t_rand;
        p.rgb_act = coarse->rgb_act; p.sigma_act = coarse->sigma_act; p.white_bkgd = cfg->white_bkgd;
        p.z_out = out->z_vals; p.rgb_map = out->rgb; p.depth_map = out->depth; p.acc_map = out->acc; p.weights = out->weights;
        return nerf_mlp_c_dispatch(coarse->W, coarse->D, coarse->skip, p, as_stream(stream));
    }
    if (!z_done && (rc = evd_sample_z(cfg, rb, nc, R, t_rand, zc, stream))) return rc;
    auto pass = [&](const evd_nerf* net, const float* z, int Sp, const float* noise, float* rgb, float* depth, float* acc,
                    float* weights, float* raw_out, float* feat) -> int {
        int rc2 = evd_nerf_mlp(net, cfg->precision, rb, z, R, Sp, raw_out, feat, out->feature_kind ? out->feature_kind : 1, stream);
        if (rc2) return rc2;
        const float thr = (!cfg->is_train && net->rmnear > 0.f) ? (float)((double)net->rmnear / 128.0) : 0.f;
        return evd_raw2outputs(raw_out, z, rb + 3, nc, R, Sp, 4, 3, 0, 3, net->rgb_act, net->sigma_act, cfg->white_bkgd, thr,
                               noise, rgb, nullptr, acc, weights, depth, nullptr, 0, nullptr, stream);
    };
    if (!Ni) {
        float* wo = out->weights ? out->weights : wts;
        float* ro = out->raw ? out->raw : raw;
        return pass(coarse, zc, S, noise0, out->rgb, out->depth, out->acc, wo, ro, out->feature);
    }
    float* w0 = out->weights0 ? out->weights0 : wts0;
    if ((rc = pass(coarse, zc, S, noise0, out->rgb0, out->depth0, out->acc0, w0, raw, nullptr))) return rc;
    float* zm = out->z_vals ? out->z_vals : z2;
    if ((rc = evd_sample_pdf_merge(zc, w0, R, S, Ni, cfg->perturb == 0.f, u, zs, zm, nullptr, out->z_std, stream))) return rc;
    float* wo = out->weights ? out->weights : wts;
    float* ro = out->raw ? out->raw : raw;
    return pass(fine, zm, St, noise1, out->rgb, out->depth, out->acc, wo, ro, out->feature);
}

extern "C" {
int evd_nerf_render_rays(const evd_nerf* coarse, const evd_nerf* fine, const evd_render_cfg* cfg, const float* rb,
                         long R, const float* t_rand, const float* u, const float* noise0, const float* noise1,
                         evd_render_out* out, void* workspace, size_t workspace_bytes, void* stream) {
    return render_rays_impl(coarse, fine, cfg, rb, R, t_rand, u, noise0, noise1, out, workspace, workspace_bytes, stream, false);
}

int evd_nerf_render(const evd_nerf* coarse, const evd_nerf* fine, const evd_render_cfg* cfg, const float* rays, long R,
                    const float* t_rand, const float* u, const float* noise0, const float* noise1,
                    evd_render_out* out, void* workspace, size_t workspace_bytes, void* stream) {
    EVD_REQUIRE(cfg && out && (rays || R == 0), "evd_nerf_render: null argument");
    if (R == 0) return EVD_OK;
    const size_t need = evd_nerf_render_workspace_bytes(cfg, R);
    if (!workspace || workspace_bytes < need)
        return fail(EVD_E_WORKSPACE, "evd_nerf_render: workspace %zu < %zu bytes", workspace_bytes, need);
    float* rb = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    // ray packing and z stratification in ONE launch (they are 44 B per ray and 4 B per sample), unless the step-fusing kernel wants z itself
    const bool fused_z = cfg->use_viewdirs && cfg->N_samples >= 1 && !env_flag("EVD_FUSE_STEP");
    int rc = fused_z ? evd_ray_batch_z(cfg, rays, R, t_rand, rb, render_z_slot(cfg, R, out, workspace), stream) : evd_ray_batch(cfg, rays, R, rb, stream);
    if (rc) return rc;
    return render_rays_impl(coarse, fine, cfg, rb, R, t_rand, u, noise0, noise1, out, workspace, workspace_bytes, stream, fused_z);
}

}  // extern "C"
