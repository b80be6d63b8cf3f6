// Backward of one PDRF level's sigma / colour networks (reference: autograd of VoxelNeRFBase.forward, networks/pdrf/voxnerf.py:210-254)
// on the dgrad / wgrad kernels of nerf_train_kernel.h; slots: voxel_mlp_kernel.h VStore.
//
//   raw = (sigma, sigmoid(colour)):  d colour_pre = d raw_c . s (1 - s) is folded into the gradient fragments
//   C2^T -> d c1 . [c1 > 0];  C1^T -> d c0 . [c0 > 0];  C0^T (geo rows) -> d geo;  [Geo^T | Sigma^T] -> d hid . [hid > 0];
//   L0^T (feature rows) -> d fts, written out as float32 rows for the tri-plane scatter (kernel_voxel.hip k_voxel_sample_bwd)
#pragma once

#include "nerf_train_kernel.h"
#include "voxel_mlp_kernel.h"
#include "voxel_train.h"
#include "voxel_bwd_fused64.h"

namespace evd {

template <int PREC>
__global__ __launch_bounds__(256) void k_voxel_grad_frags(const float* __restrict__ d_raw, const float* __restrict__ raw, long nsamp,
                                                          const unsigned* __restrict__ maxbits, char* __restrict__ store, long tiles,
                                                          long tile_bytes, int g_col, int g_sig) {
    typedef POps<PREC> O;
    typedef typename O::B B;
    pipe_fp16_saturate<PREC>();
    const long idx = (long)blockIdx.x * 256 + threadIdx.x, tile = idx >> 6;
    if (tile >= tiles) return;
    const int lane = idx & 63, n = lane & 31, h = lane >> 5;
    const long smp = tile * 32 + n;
    const float s = grad_scale(*maxbits, false);
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    if (h == 0 && smp < nsamp) {
        g = *reinterpret_cast<const f32x4*>(d_raw + smp * 4);
        const f32x4 r = *reinterpret_cast<const f32x4*>(raw + smp * 4);
#pragma unroll
        for (int c = 1; c < 4; ++c) g[c] = g[c] * r[c] * (1.f - r[c]);         // through torch.sigmoid (voxnerf.py:252)
    }
    B col, sg;
    O::zero(col);
    O::zero(sg);
    O::template set_pair<false>(col, 0, g[1] * s, g[2] * s);
    O::template set_pair<false>(col, 1, g[3] * s, 0.f);
    O::template set_pair<false>(sg, 0, g[0] * s, 0.f);
    constexpr int FB = frag_bytes(PREC);
    char* a = store + tile * tile_bytes + lane * 16;
    act_store<FB>(a, g_col, col);
    act_store<FB>(a, g_sig, sg);
}

// gradient fragments of the input features -> float32 rows [n, FT] (loss scale removed): fragment j, position kk of the dgrad
// output <-> feature 16 j + phi(kk)
template <int PREC>
__global__ __launch_bounds__(256) void k_frags_to_rows(const char* __restrict__ store, long tile_bytes, int slot, int nfrag, long nsamp,
                                                       const unsigned* __restrict__ maxbits, float* __restrict__ rows, int stride) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long tile = idx / (64 * nfrag);
    const int j = (int)((idx / 64) % nfrag), lane = idx & 63, n = lane & 31, h = lane >> 5;
    const long smp = tile * 32 + n;
    if (smp >= nsamp) return;
    float fv[8];
    frag_values<PREC>(store + tile * tile_bytes + lane * 16, slot + j, fv);
    const float inv = grad_scale(*maxbits, true);
#pragma unroll
    for (int e = 0; e < 8; ++e) rows[smp * (long)stride + 16 * j + phi(8 * h + e)] = fv[e] * inv;
}

// float32 gradient rows (the per-sample geo features' gradient, from the AWP consumer) added into gradient fragments: fragment j,
// position kk <-> channel 16 j + phi(kk); rows are scaled by the loss scale first
template <int PREC>
__global__ __launch_bounds__(256) void k_rows_add_to_frags(char* __restrict__ store, long tile_bytes, int slot, int nfrag, long nsamp,
                                                           const unsigned* __restrict__ maxbits, const float* __restrict__ rows, int stride) {
    typedef POps<PREC> O;
    pipe_fp16_saturate<PREC>();
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long tile = idx / (64 * nfrag);
    const int j = (int)((idx / 64) % nfrag), lane = idx & 63, n = lane & 31, h = lane >> 5;
    const long smp = tile * 32 + n;
    if (smp >= nsamp) return;
    char* a = store + tile * tile_bytes + lane * 16;
    const float s = grad_scale(*maxbits, false);
    float v[8];
    frag_values<PREC>(a, slot + j, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += rows[smp * (long)stride + 16 * j + phi(8 * h + e)] * s;
    typename O::B b;
#pragma unroll
    for (int e = 0; e < 4; ++e) O::template set_pair<false>(b, e, v[2 * e], v[2 * e + 1]);
    act_store<frag_bytes(PREC)>(a, slot + j, b);
}

// *dst = max(*dst, *src) on float bits (both non-negative): the loss scale also covers a gradient that arrives as fragments
static __global__ void k_max_word(unsigned* __restrict__ dst, const unsigned* __restrict__ src) { atomicMax(dst, *src); }

// gradient fragments of ANOTHER store (the AWP embedding's d geo, awp_embed.h: its own loss-scale word) added into this level's
template <int PREC>
__global__ __launch_bounds__(256) void k_frags_add_scaled(char* __restrict__ store, long tile_bytes, int slot, const char* __restrict__ src, long src_tile_bytes,
                                                          int src_slot, int nfrag, long tiles, const unsigned* __restrict__ maxbits,
                                                          const unsigned* __restrict__ src_maxbits) {
    typedef POps<PREC> O;
    pipe_fp16_saturate<PREC>();
    const long idx = (long)blockIdx.x * 256 + threadIdx.x, tile = idx / (64 * nfrag);
    if (tile >= tiles) return;
    const int j = (int)((idx / 64) % nfrag), lane = idx & 63;
    char* a = store + tile * tile_bytes + lane * 16;
    static_assert(is_half_prec(PREC), "the AWP embedding's store is half precision: both stores share one fragment format");
    const float s = grad_scale(*maxbits, false) * grad_scale(*src_maxbits, true);
    float v[8], g[8];
    frag_values<PREC>(a, slot + j, v);
    frag_values<PREC>(src + tile * src_tile_bytes + lane * 16, src_slot + j, g);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaf(g[e], s, v[e]);
    typename O::B b;
#pragma unroll
    for (int e = 0; e < 4; ++e) O::template set_pair<false>(b, e, v[2 * e], v[2 * e + 1]);
    act_store(a, slot + j, b);
}

template <int PREC, int HD, int G, int FT> static int run_voxel_backward(const VoxBwdPlan& b, hipStream_t st) {
    typedef VStore<HD, G, FT> VS;
    constexpr int T = HD / 32, KS = HD / 16, KF = FT / 16, GT = VS::GT, FTT = (FT + 31) / 32, IC = 3 * (1 + 2 * PE_L), ICV = 3 * (1 + 2 * PE_LV);
    int rc;
    EVD_HIP(hipMemsetAsync(b.maxbits, 0, sizeof(unsigned), st));
    hipLaunchKernelGGL(k_absmax, dim3(512), dim3(256), 0, st, b.d_raw, b.nsamp * 4, b.maxbits);
    EVD_LAUNCH_CHECK();
    if (b.d_feature) {          // the loss scale covers both incoming gradients
        hipLaunchKernelGGL(k_absmax, dim3(512), dim3(256), 0, st, b.d_feature, b.nsamp * G, b.maxbits);
        EVD_LAUNCH_CHECK();
    }
    if (b.awp_store) {          // ... and the d geo fragments of the AWP embedding's backward (true-unit maximum in its trailer)
        hipLaunchKernelGGL(k_max_word, dim3(1), dim3(1), 0, st, b.maxbits, b.awp_words + 1);
        EVD_LAUNCH_CHECK();
    }
    // the 64-wide level: the whole chain in one launch, the tile's gradient resident in registers (voxel_bwd_fused64.h); EVD_BWD_FUSE64=0
    // keeps the per-layer chain below (A/B, and the reference the fused kernel is tested against)
    if constexpr (is_half_prec(PREC) && HD == 64 && G == 15 && FT == 32) {
        static const bool fuse64 = [] { const char* e = getenv("EVD_BWD_FUSE64"); return !(e && e[0] == '0'); }();
        if (fuse64 && !b.d_feature && !b.awp_store) {
            const int blocks = (int)(cdiv(b.tiles, 4L) < 256 ? cdiv(b.tiles, 4L) : 256);
            VoxBwdFusedParams fp;
            fp.d_raw = b.d_raw; fp.raw = b.raw; fp.nsamp = b.nsamp; fp.tiles = b.tiles; fp.store = b.store; fp.maxbits = b.maxbits; fp.partial = b.partial;
            for (int k = 0; k < VBWD_NSTREAMS; ++k) fp.wt[k] = b.wt[k];
            // d fts straight as rows when they can be written with 16-byte stores (EVD_BWD_ROWS=0: fragments + k_frags_to_rows)
            static const bool rows_on = [] { const char* e = getenv("EVD_BWD_ROWS"); return !(e && e[0] == '0'); }();
            const bool rows_direct = rows_on && b.d_fts && b.d_fts_stride % 4 == 0 && ((uintptr_t)b.d_fts & 15) == 0;
            fp.d_fts = rows_direct ? b.d_fts : nullptr; fp.d_fts_stride = b.d_fts_stride;
            EVD_SET_MAX_LDS((&k_voxel_bwd_fused64<PREC>), (size_t)f64::LDS_BYTES);
            hipLaunchKernelGGL((k_voxel_bwd_fused64<PREC>), dim3((unsigned)blocks), dim3(256), (size_t)f64::LDS_BYTES, st, fp);
            EVD_LAUNCH_CHECK();
            const VoxBwdGrads& g = b.grads;
            WreduceJobs jobs;
            auto job = [&](int i, int a0, int RT, int CT, bool bias, int ymap, int xmap, float* dW, int ld, float* db) {
                WreduceParams& q = jobs.j[i];
                q.partial = b.partial + (long)a0 * 1024; q.nparts = blocks; q.RT = dW ? RT : 0; q.CT = CT; q.NC = CT + (bias ? 1 : 0);
                q.rowmap = b.maps + ymap; q.colmap = b.maps + xmap; q.dW = dW; q.ld = ld; q.db = bias ? db : nullptr; q.maxbits = b.maxbits;
                q.accum = b.accumulate; q.part_stride = (long)f64::NBLK * 1024;
            };
            job(0, f64::A_C2, 1, 2, false, VMAP_COL, VMAP_HID, g.color_w[2], HD, nullptr);
            job(1, f64::A_C1, 2, 2, false, VMAP_HID, VMAP_HID, g.color_w[1], HD, nullptr);
            job(2, f64::A_C0, 2, 2, false, VMAP_HID, VMAP_F64_C0, g.color_w[0], G + ICV, nullptr);
            job(3, f64::A_SG, 1, 2, false, VMAP_F64_SG, VMAP_HID, g.sigma_w[1], HD, nullptr);
            job(4, f64::A_L0, 2, 3, false, VMAP_HID, VMAP_F64_L0, g.sigma_w[0], FT + IC, nullptr);
            hipLaunchKernelGGL(k_wgrad_reduce_jobs, dim3(2 * 3 * 4, WREDUCE_MAX_JOBS), dim3(256), 0, st, jobs);
            EVD_LAUNCH_CHECK();
            if (g.color_b[0] || g.color_b[1] || g.color_b[2]) {        // the shared bias block (columns: colour_net.2, .1 x 2 row tiles, .0 x 2)
                BiasColsParams bp;
                bp.partial = b.partial + (long)f64::A_BIAS * 1024; bp.nparts = blocks; bp.part_stride = (long)f64::NBLK * 1024; bp.ncols = 5;
                bp.rowmap[f64::B_C2] = b.maps + VMAP_COL; bp.db[f64::B_C2] = g.color_b[2];
                for (int yb = 0; yb < 2; ++yb) {
                    bp.rowmap[f64::B_C1 + yb] = b.maps + VMAP_HID + 32 * yb; bp.db[f64::B_C1 + yb] = g.color_b[1];
                    bp.rowmap[f64::B_C0 + yb] = b.maps + VMAP_HID + 32 * yb; bp.db[f64::B_C0 + yb] = g.color_b[0];
                }
                bp.maxbits = b.maxbits; bp.accum = b.accumulate;
                hipLaunchKernelGGL(k_bias_cols_reduce, dim3(5 * 32 / 4), dim3(256), 0, st, bp);
                EVD_LAUNCH_CHECK();
            }
            if (b.d_dirs) {
                hipLaunchKernelGGL((k_pe_bwd<PREC, PE_LV, PEV_KS>), dim3((unsigned)cdiv(b.tiles * 64, 256L)), dim3(256), 0, st, (const char*)b.store, VS::tile_bytes(PREC),
                                   VS::D_DIRPE, b.nsamp, b.viewdirs, b.vd_stride, b.S, b.maxbits, b.d_dirs, 0);
                EVD_LAUNCH_CHECK();
            }
            if (b.d_fts && !rows_direct) {
                hipLaunchKernelGGL((k_frags_to_rows<PREC>), dim3((unsigned)cdiv(b.tiles * 64 * KF, 256L)), dim3(256), 0, st, (const char*)b.store, VS::tile_bytes(PREC),
                                   VS::D_FTS, KF, b.nsamp, b.maxbits, b.d_fts, b.d_fts_stride);
                EVD_LAUNCH_CHECK();
            }
            if (b.d_pts) {
                hipLaunchKernelGGL((k_pe_bwd<PREC, PE_L, PE_KS>), dim3((unsigned)cdiv(b.tiles * 64, 256L)), dim3(256), 0, st, (const char*)b.store, VS::tile_bytes(PREC),
                                   VS::D_PE, b.nsamp, b.pts, 3, 1, b.maxbits, b.d_pts, 0);
                EVD_LAUNCH_CHECK();
            }
            return EVD_OK;
        }
    }
    hipLaunchKernelGGL((k_voxel_grad_frags<PREC>), dim3((unsigned)cdiv(b.tiles * 64, 256L)), dim3(256), 0, st, b.d_raw, b.raw, b.nsamp, b.maxbits, b.store,
                       b.tiles, VS::tile_bytes(PREC), VS::G_COL, VS::G_SIG);
    EVD_LAUNCH_CHECK();
    auto dgrad = [&](int stream, int in_slot, int extra_slot, int mask_slot, int out_slot) {
        DgradParams p;
        p.wstream = b.wt[stream]; p.store = b.store; p.tile_bytes = VS::tile_bytes(PREC);
        p.in_slot = in_slot; p.extra_slot = extra_slot; p.mask_slot = mask_slot; p.out_slot = out_slot;
        return p;
    };
    auto wgrad = [&](auto launch, int RT, int CT, bool bias, int y_slot, int x_slot, int ymap, int xmap, float* dW, int ld, float* db) -> int {
        if (!dW) return EVD_OK;
        const int blocks = (int)(cdiv(b.tiles, (long)WGRAD_TPI) < b.wgrad_blocks ? cdiv(b.tiles, (long)WGRAD_TPI) : b.wgrad_blocks);
        WgradParams p;
        p.store = b.store; p.tiles = b.tiles; p.tile_bytes = VS::tile_bytes(PREC); p.y_slot = y_slot; p.x_slot = x_slot; p.bias = bias ? 1 : 0; p.partial = b.partial;
        hipStream_t ws = st;
        if (b.side) {                           // fork: everything issued so far on the caller's stream first (nerf_train_kernel.h)
            EVD_HIP(hipEventRecord(b.ev, st));
            EVD_HIP(hipStreamWaitEvent(b.side, b.ev, 0));
            ws = b.side;
            if (int rcs = test_side_spin(ws)) return rcs;
        }
        int r = launch(p, blocks, ws);
        if (r) return r;
        WreduceParams q;
        q.partial = b.partial; q.nparts = blocks; q.RT = RT; q.CT = CT; q.NC = CT + (bias ? 1 : 0);
        q.rowmap = b.maps + ymap; q.colmap = b.maps + xmap; q.dW = dW; q.ld = ld; q.db = bias ? db : nullptr; q.maxbits = b.maxbits; q.accum = b.accumulate;
        hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((long)RT * q.NC * 4)), dim3(256), 0, ws, q);
        EVD_LAUNCH_CHECK();
        return EVD_OK;
    };
    // wgrad(l) with dgrad(l) in one launch (nerf_train_kernel.h k_wgrad_dgrad): the 256-wide layers of the fine level in the
    // half-precision modes; EVD_BWD_FUSE=0 keeps the separate launches (A/B)
    static const bool fuse_on = [] { const char* e = getenv("EVD_BWD_FUSE"); return !(e && e[0] == '0'); }();
    constexpr bool FUSABLE = is_half_prec(PREC) && T == 8;
    auto fused = [&](auto launch, int CT, bool bias, int y_slot, int x_slot, int ymap, int xmap, float* dW, int ld, float* db, int stream, int mask_slot, int out_slot,
                     int RTr = 8, int y_last_slot = -1, const char* ygen_wt = nullptr, float* rows = nullptr, int rows_tiles = 0) -> int {
        const int blocks = (int)(b.tiles < b.wgrad_blocks ? b.tiles : b.wgrad_blocks);
        WgradFusedParams p;
        p.w.store = b.store; p.w.tiles = b.tiles; p.w.tile_bytes = VS::tile_bytes(PREC); p.w.y_slot = y_slot; p.w.x_slot = x_slot; p.w.bias = bias ? 1 : 0; p.w.partial = b.partial;
        p.wt = b.wt[stream]; p.out_store = b.store; p.mask_slot = mask_slot; p.out_slot = out_slot; p.y_last_slot = y_last_slot; p.ygen_wt = ygen_wt;
        p.rows = rows; p.rows_stride = b.d_fts_stride; p.rows_tiles = rows_tiles; p.nsamp = b.nsamp; p.maxbits = b.maxbits;
        if (b.side && !test_skip_side_join()) { // the wgrad launches in flight on the side stream use the partial scratch: join first
            EVD_HIP(hipEventRecord(b.ev, b.side));
            EVD_HIP(hipStreamWaitEvent(st, b.ev, 0));
        }
        int r = launch(p, blocks, st);
        if (r) return r;
        if (!dW) return EVD_OK;
        WreduceParams q;
        q.partial = b.partial; q.nparts = blocks; q.RT = RTr; q.CT = CT; q.NC = CT + (bias ? 1 : 0);
        q.rowmap = b.maps + ymap; q.colmap = b.maps + xmap; q.dW = dW; q.ld = ld; q.db = bias ? db : nullptr; q.maxbits = b.maxbits; q.accum = b.accumulate;
        q.part_stride = RTr < 8 ? (long)8 * q.NC * 1024 : 0;       // the kernel lays every workgroup's set out as 8 row tiles
        hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((long)RTr * q.NC * 4)), dim3(256), 0, st, q);
        EVD_LAUNCH_CHECK();
        return EVD_OK;
    };
    const VoxBwdGrads& g = b.grads;
    // (each wgrad is issued before the dgrad layer that reads the same arrays: independent, concurrent on the side stream)
    // color_net.2 (+ sigmoid, folded into the gradient fragment)
    if ((rc = wgrad(launch_wgrad<PREC, 1, T, true>, 1, T, g.color_b[2] != nullptr, VS::G_COL, VS::C1, VMAP_COL, VMAP_HID, g.color_w[2], HD, g.color_b[2]))) return rc;
    // color_net.1; in the fused form d c1 = (W2^T d colour) . [c1 > 0] is formed inside the launch from the pair [G_COL | M_C1]
    // (k_wgrad_dgrad YGEN: color_net.2's dgrad launch and the D_C1 round trip are gone; EVD_BWD_YGEN=0: the separate launch)
    static const bool ygen_on = [] { const char* e = getenv("EVD_BWD_YGEN"); return !(e && e[0] == '0'); }();
    const bool ygen = FUSABLE && fuse_on && ygen_on && g.color_w[1];
    static_assert(VS::M_C1 == VS::G_COL + 1, "the formed gradient's two inputs are one fragment pair");
    if (!ygen && (rc = launch_dgrad<PREC, 1, T, 1, false, 2>(dgrad(VBWD_C2, VS::G_COL, -1, VS::M_C1, VS::D_C1), b.tiles, st))) return rc;
    if constexpr (FUSABLE) {
        if (ygen) {
            if ((rc = fused(launch_wgrad_dgrad<PREC, 8, 8, 1, 8, 16, true>, 8, g.color_b[1] != nullptr, VS::G_COL, VS::C1 - KS, VMAP_HID, VMAP_HID, g.color_w[1], HD, g.color_b[1], VBWD_C1, VS::M_C0,
                            VS::D_C0, 8, -1, b.wt[VBWD_C2]))) return rc;
        } else if (fuse_on && g.color_w[1]) {
            if ((rc = fused(launch_wgrad_dgrad<PREC, 8, 8, 1>, 8, g.color_b[1] != nullptr, VS::D_C1, VS::C1 - KS, VMAP_HID, VMAP_HID, g.color_w[1], HD, g.color_b[1], VBWD_C1, VS::M_C0, VS::D_C0))) return rc;
        } else {
            if ((rc = wgrad(launch_wgrad<PREC, T, T, false>, T, T, g.color_b[1] != nullptr, VS::D_C1, VS::C1 - KS, VMAP_HID, VMAP_HID, g.color_w[1], HD, g.color_b[1]))) return rc;
            if ((rc = launch_dgrad<PREC, KS, T, KS, false, 2>(dgrad(VBWD_C1, VS::D_C1, -1, VS::M_C0, VS::D_C0), b.tiles, st))) return rc;
        }
    } else {
        if ((rc = wgrad(launch_wgrad<PREC, T, T, false>, T, T, g.color_b[1] != nullptr, VS::D_C1, VS::C1 - KS, VMAP_HID, VMAP_HID, g.color_w[1], HD, g.color_b[1]))) return rc;
        if ((rc = launch_dgrad<PREC, KS, T, KS, false, 2>(dgrad(VBWD_C1, VS::D_C1, -1, VS::M_C0, VS::D_C0), b.tiles, st))) return rc;
    }
    // color_net.0 on cat([geo, PE(dirs)])
    bool c0_fused = false;
    if constexpr (FUSABLE && G == 32 * GT) {
        if (fuse_on && g.color_w[0]) {     // wgrad + dgrad of color_net.0 in one launch: d c0 read once; writes d geo | d PE(dirs) (no ReLU on geo)
            static_assert(VS::D_DIRPE == VS::D_GEO + 2 * GT, "d PE(dirs) behind d geo");
            if ((rc = fused(launch_wgrad_dgrad<PREC, GT + 1, GT + 1, 0>, GT + 1, g.color_b[0] != nullptr, VS::D_C0, VS::GEO, VMAP_HID, VMAP_GEO_X, g.color_w[0], G + ICV, g.color_b[0], VBWD_C0, -1, VS::D_GEO))) return rc;
            c0_fused = true;
        }
    }
    if (c0_fused) {
    } else if constexpr (is_half_prec(PREC) && G == 32 * GT) {
        // geo and direction-encoding columns in ONE launch (adjacent fragments, adjacent index maps): d c0 is read once
        static_assert(VS::DIRPE == VS::GEO + 2 * GT && VMAP_DIR == VMAP_GEO_X + 128 && (G == 128), "adjacent fragments and column maps");
        if ((rc = wgrad(launch_wgrad<PREC, T, GT + 1, false>, T, GT + 1, g.color_b[0] != nullptr, VS::D_C0, VS::GEO, VMAP_HID, VMAP_GEO_X, g.color_w[0], G + ICV, g.color_b[0]))) return rc;
    } else {
        if ((rc = wgrad(launch_wgrad<PREC, T, GT, false>, T, GT, g.color_b[0] != nullptr, VS::D_C0, VS::GEO, VMAP_HID, VMAP_GEO_X, g.color_w[0], G + ICV, g.color_b[0]))) return rc;
        if ((rc = wgrad(launch_wgrad<PREC, T, 1, false>, T, 1, false, VS::D_C0, VS::DIRPE, VMAP_HID, VMAP_DIR, g.color_w[0], G + ICV, nullptr))) return rc;
    }
    if (!c0_fused && (rc = launch_dgrad<PREC, KS, GT + 1, KS, false, 0>(dgrad(VBWD_C0, VS::D_C0, -1, -1, VS::D_GEO), b.tiles, st))) return rc;   // d geo | d PE(dirs)
    if (b.d_feature) {          // + the gradient of the geo features as an output of the level (voxnerf.py:221, consumed by AWP)
        if (G % 16) return fail(EVD_E_INVALID, "evd_voxel_mlp_backward: d_feature is built for the fine level (geo 128)");
        hipLaunchKernelGGL((k_rows_add_to_frags<PREC>), dim3((unsigned)cdiv(b.tiles * 64 * (G / 16), 256L)), dim3(256), 0, st, b.store, VS::tile_bytes(PREC), VS::D_GEO,
                           G / 16, b.nsamp, b.maxbits, b.d_feature, G);
        EVD_LAUNCH_CHECK();
    }
    if (b.awp_store) {          // + the gradient that reached the geo features through the fused AWP embedding (awp_embed_kernel.h)
        if (G % 16) return fail(EVD_E_INVALID, "evd_voxel_mlp_backward: the AWP embedding reads the fine level (geo 128)");
        if constexpr (!is_half_prec(PREC)) return fail(EVD_E_INVALID, "evd_voxel_mlp_backward: awp_store goes with the f16 / bf16 modes (pass d_feature rows in f16x3)");
        else hipLaunchKernelGGL((k_frags_add_scaled<PREC>), dim3((unsigned)cdiv(b.tiles * 64 * (G / 16), 256L)), dim3(256), 0, st, b.store, VS::tile_bytes(PREC), VS::D_GEO,
                           b.awp_store, b.awp_tile_bytes, b.awp_slot, G / 16, b.tiles, b.maxbits, b.awp_words);
        EVD_LAUNCH_CHECK();
    }
    if (b.d_dirs) {
        hipLaunchKernelGGL((k_pe_bwd<PREC, PE_LV, PEV_KS>), dim3((unsigned)cdiv(b.tiles * 64, 256L)), dim3(256), 0, st, (const char*)b.store, VS::tile_bytes(PREC),
                           VS::D_DIRPE, b.nsamp, b.viewdirs, b.vd_stride, b.S, b.maxbits, b.d_dirs, 0);
        EVD_LAUNCH_CHECK();
    }
    // sigma_net.1 = [sigma row | geo rows] on hid
    bool sg_fused = false;
    if constexpr (FUSABLE && GT == 4) {
        // round 5: both wgrads and the dgrad in one launch -- the gradient as 4 geo row tiles + the d sigma fragment (k_wgrad_dgrad RT_ = 5,
        // 9 k-steps): hid and d geo are read once instead of three / two times (42 KiB per tile instead of 67); EVD_BWD_FUSE_SG=0: the three launches
        static const bool sg_on = [] { const char* e = getenv("EVD_BWD_FUSE_SG"); return !(e && e[0] == '0'); }();
        if (fuse_on && sg_on && g.sigma_w[1]) {
            if ((rc = fused(launch_wgrad_dgrad<PREC, 8, 8, 1, GT + 1, 2 * GT + 1>, T, false, VS::D_GEO, VS::HID, VMAP_SG5, VMAP_HID, g.sigma_w[1], HD, nullptr, VBWD_SIGGEO,
                            VS::M_HID, VS::D_HID, GT + 1, VS::G_SIG))) return rc;
            sg_fused = true;
        }
    }
    if (!sg_fused) {
        if ((rc = wgrad(launch_wgrad<PREC, GT, T, false>, GT, T, false, VS::D_GEO, VS::HID, VMAP_GEO_Y, VMAP_HID, g.sigma_w[1], HD, nullptr))) return rc;
        if ((rc = wgrad(launch_wgrad<PREC, 1, T, true>, 1, T, false, VS::G_SIG, VS::HID, VMAP_SIG, VMAP_HID, g.sigma_w[1], HD, nullptr))) return rc;
        if ((rc = launch_dgrad<PREC, 2 * GT + 1, T, 2 * GT, true, 2>(dgrad(VBWD_SIGGEO, VS::D_GEO, VS::G_SIG, VS::M_HID, VS::D_HID), b.tiles, st))) return rc;
    }
    // sigma_net.0 on cat([fts, PE(pts)])
    bool l0_fused = false, l0_rows = false;
    if constexpr (FUSABLE && FTT == 2) {
        if (fuse_on && g.sigma_w[0] && (b.d_fts || b.d_pts)) {     // ... with its dgrad (d fts | d PE(pts)) in one launch
            static_assert(VS::D_PE == VS::D_FTS + 2 * FTT, "d PE(pts) behind d fts");
            // round 5: the feature tiles of d X leave as the float32 rows the scatter reads (EVD_BWD_ROWS=0: fragments + k_frags_to_rows)
            static const bool rows_on = [] { const char* e = getenv("EVD_BWD_ROWS"); return !(e && e[0] == '0'); }();
            l0_rows = rows_on && b.d_fts && b.d_fts_stride % 4 == 0 && ((uintptr_t)b.d_fts & 15) == 0;
            if (l0_rows) rc = fused(launch_wgrad_dgrad<PREC, FTT + 2, FTT + 2, 0, 8, 16, false, true>, FTT + 2, false, VS::D_HID, VS::IN0, VMAP_HID, VMAP_FTS, g.sigma_w[0], FT + IC, nullptr,
                                    VBWD_L0, -1, VS::D_FTS, 8, -1, nullptr, b.d_fts, FTT);
            else rc = fused(launch_wgrad_dgrad<PREC, FTT + 2, FTT + 2, 0>, FTT + 2, false, VS::D_HID, VS::IN0, VMAP_HID, VMAP_FTS, g.sigma_w[0], FT + IC, nullptr, VBWD_L0, -1, VS::D_FTS);
            if (rc) return rc;
            l0_fused = true;
        }
    }
    if (l0_fused) {
    } else if constexpr (FTT == 2) {
        // the feature and encoding columns in ONE launch (their fragments and their index maps are adjacent): d hid is read once, not twice
        static_assert(VMAP_PE == VMAP_FTS + 32 * FTT, "adjacent column maps");
        if ((rc = wgrad(launch_wgrad<PREC, T, FTT + 2, false>, T, FTT + 2, false, VS::D_HID, VS::IN0, VMAP_HID, VMAP_FTS, g.sigma_w[0], FT + IC, nullptr))) return rc;
    } else {
        if ((rc = wgrad(launch_wgrad<PREC, T, FTT, false>, T, FTT, false, VS::D_HID, VS::IN0, VMAP_HID, VMAP_FTS, g.sigma_w[0], FT + IC, nullptr))) return rc;
        if ((rc = wgrad(launch_wgrad<PREC, T, 2, false>, T, 2, false, VS::D_HID, VS::IN0 + KF, VMAP_HID, VMAP_PE, g.sigma_w[0], FT + IC, nullptr))) return rc;
    }
    if (b.d_fts || b.d_pts) {
        if (!l0_fused && (rc = launch_dgrad<PREC, KS, FTT + 2, KS, false, 0>(dgrad(VBWD_L0, VS::D_HID, -1, -1, VS::D_FTS), b.tiles, st))) return rc;       // d fts | d PE(pts)
        if (b.d_fts && !l0_rows) {
            hipLaunchKernelGGL((k_frags_to_rows<PREC>), dim3((unsigned)cdiv(b.tiles * 64 * KF, 256L)), dim3(256), 0, st, (const char*)b.store, VS::tile_bytes(PREC),
                               VS::D_FTS, KF, b.nsamp, b.maxbits, b.d_fts, b.d_fts_stride);
            EVD_LAUNCH_CHECK();
        }
        if (b.d_pts) {
            hipLaunchKernelGGL((k_pe_bwd<PREC, PE_L, PE_KS>), dim3((unsigned)cdiv(b.tiles * 64, 256L)), dim3(256), 0, st, (const char*)b.store, VS::tile_bytes(PREC),
                               VS::D_PE, b.nsamp, b.pts, 3, 1, b.maxbits, b.d_pts, 0);
            EVD_LAUNCH_CHECK();
        }
    }
    if (b.side && !test_skip_side_join()) {     // join
        EVD_HIP(hipEventRecord(b.ev, b.side));
        EVD_HIP(hipStreamWaitEvent(st, b.ev, 0));
    }
    return EVD_OK;
}

}  // namespace evd
