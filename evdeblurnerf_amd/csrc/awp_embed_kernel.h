// Kernels of the AWP sample-feature embedding (awp_embed.h): forward on the software pipeline (one straight-line stream of 40 MFMAs
// per wavefront), backward on the dgrad / wgrad kernels of nerf_train_kernel.h with the 64-wide layer table below.
// reference: networks/dpnerf/awp.py:98-102 (forward), its autograd during run_nerf.py:593-601 (backward).
#pragma once

#include "awp_embed.h"
#include "voxel_train_kernel.h"
#include "awp_bwd_fused.h"

namespace evd {

template <class C, bool TRAIN> struct AwpNet {
    static constexpr int KW = AWP_W / 16, KIN = AWP_IN / 16, T = AWP_W / 32, PD = C::PD;
    static constexpr int slot(int s) { return TRAIN ? s : -1; }
    // layer l: k-steps in the hidden arrangement 16 j + phi(kk); the last tile of a layer stays pending in the accumulators and is
    // drained by the next layer into its k-steps 2, 3 (PDOFF = 2)
    typedef LayerDesc<KIN, T, 1, true, false, 0, 0, false, 0, 0, 0, false, 0, -1, false, 1, slot(awpstore::E0)> E0;
    static constexpr int F1 = T * KIN;
    typedef LayerDesc<KW, T, 1, true, false, 0, F1, false, F1 % PD, E0::PAR_OUT, 1, true, 2 * (T - 1), -1, false, 1, slot(awpstore::E0 + KW),
                      slot(awpstore::E0 + 2 * (T - 1))> E1;
    static constexpr int F2 = F1 + T * KW;
    typedef LayerDesc<KW, T, 1, true, false, 0, F2, false, F2 % PD, E1::PAR_OUT, 1, true, 2 * (T - 1), -1, false, 1, slot(awpstore::E0 + 2 * KW),
                      slot(awpstore::E0 + KW + 2 * (T - 1))> E2;
    static constexpr int F3 = F2 + T * KW;
    // the last layer also writes its output as float32 rows (h_local: what feature_integration and the MAM read)
    typedef LayerDesc<KW, T, 1, true, false, 0, F3, true, F3 % PD, E2::PAR_OUT, 1, true, 2 * (T - 1), -1, true, 0, slot(awpstore::E0 + 3 * KW),
                      slot(awpstore::E0 + 2 * KW + 2 * (T - 1))> E3;
    static constexpr int NCH = cceil(F3 + T * KW, C::FPC);
    static_assert(NCH == AWP_NCHUNKS && T == 2 && AWP_D == 4, "stream geometry (evd_awp_api.hip packs the same table)");
};

// ROWS: the geo features come as float32 rows [n, 128] (inference, tests); otherwise as the fragments of the fine level's store
template <int PREC, bool TRAIN, bool ROWS>
__global__ __launch_bounds__(512, 2) void k_awp_embed(const AwpFwdParams p) {
    typedef PipeCfg<PREC, 1, 512> C;
    typedef typename C::O O;
    typedef typename O::B B;
    typedef AwpNet<C, TRAIN> N;
    typedef PStream<C, true, N::NCH> ST;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    pipe_fp16_saturate<PREC>();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 31, h = lane >> 5;
    ST st;
    st.start_issue(p.wstream, smem, tid);
    float* bias = reinterpret_cast<float*>(smem + C::RING);
    for (int i = tid; i < AWP_D * AWP_W; i += 512) bias[i] = p.bias[i];

    const long tile = (long)blockIdx.x * 8 + wave, smp = tile * 32 + n;
    const bool valid = smp < p.nsamp;
    const long sidx = valid ? smp : p.nsamp - 1;
    char* actl[1] = {TRAIN ? p.act + tile * awpstore::TILE_BYTES + lane * 16 : nullptr};
    B in0[1][N::KIN];
    if constexpr (ROWS) {
        // B position 8 h + e of k-step j <-> channel 16 j + phi(8 h + e) = 16 j + 8 (e >> 2) + 4 h + (e & 3)
        const float* f = p.geo_rows + sidx * AWP_IN + 4 * h;
#pragma unroll
        for (int j = 0; j < N::KIN; ++j) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(f + 16 * j), b = *reinterpret_cast<const f32x4*>(f + 16 * j + 8);
            O::template set_pair<false>(in0[0][j], 0, a[0], a[1]);
            O::template set_pair<false>(in0[0][j], 1, a[2], a[3]);
            O::template set_pair<false>(in0[0][j], 2, b[0], b[1]);
            O::template set_pair<false>(in0[0][j], 3, b[2], b[3]);
        }
    } else {
        const char* g = p.geo_frags + tile * p.geo_tile_bytes + lane * 16;
#pragma unroll
        for (int j = 0; j < N::KIN; ++j) in0[0][j] = frag_load<B>(g, p.geo_slot + j);
    }
    if constexpr (TRAIN) {
#pragma unroll
        for (int j = 0; j < N::KIN; ++j) act_store(actl[0], awpstore::GEO + j, in0[0][j]);
    }
    float* frow[1] = {valid ? p.h_local + sidx * AWP_W : nullptr};
    float* nofrow[1] = {nullptr};
    st.start_wait();
    Pipe<C> pp;
    pipe_prime<C, typename N::E0>(st, pp, bias, lane);

    B e0[1][N::KW], e1[1][N::KW], e2[1][N::KW], e3[1][N::KW];
    pipe_layer<C, typename N::E0, ST, N::KW, TRAIN>(st, pp, in0, e0, nullptr, bias, lane, nofrow, actl);
    pipe_layer<C, typename N::E1, ST, N::KW, TRAIN>(st, pp, e0, e1, nullptr, bias + AWP_W, lane, nofrow, actl);
    pipe_layer<C, typename N::E2, ST, N::KW, TRAIN>(st, pp, e1, e2, nullptr, bias + 2 * AWP_W, lane, nofrow, actl);
    pipe_layer<C, typename N::E3, ST, N::KW, TRAIN>(st, pp, e2, e3, nullptr, bias + 3 * AWP_W, lane, frow, actl);
    {   // the pending last tile of E3: float32 rows, and (training) its fragments
        typedef typename N::E3 L;
        constexpr int cur = (L::PAR + L::NG - 1) & 1, f0 = 2 * (N::T - 1);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            drain_pair<C, true>(pp.acc[cur][0][0], pp.accx[cur][0][0], k, e3[0][f0 + (k >> 2)],
                                frow[0] ? frow[0] + 32 * (N::T - 1) + 8 * (k >> 1) + 4 * h : nullptr);
        if constexpr (TRAIN) {
            act_store(actl[0], awpstore::E0 + 3 * N::KW + f0, e3[0][f0]);
            act_store(actl[0], awpstore::E0 + 3 * N::KW + f0 + 1, e3[0][f0 + 1]);
        }
    }
}

template <int PREC, bool TRAIN, bool ROWS>
static int launch_awp_embed_t(const AwpFwdParams& p, hipStream_t st) {
    typedef PipeCfg<PREC, 1, 512> C;
    const size_t lds = C::RING + AWP_D * AWP_W * sizeof(float);
    const long blocks = TRAIN ? awp_tiles(p.nsamp) / 8 : cdiv(p.nsamp, 256L);
    EVD_SET_MAX_LDS((&k_awp_embed<PREC, TRAIN, ROWS>), lds);
    hipLaunchKernelGGL((k_awp_embed<PREC, TRAIN, ROWS>), dim3((unsigned)blocks), dim3(512), lds, st, p);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

template <int PREC> static int launch_awp_embed(bool train, const AwpFwdParams& p, hipStream_t st) {
    if (p.nchunks != AWP_NCHUNKS) return fail(EVD_E_INVALID, "evd_awp_embed: packed stream has %d chunks, kernel expects %d", p.nchunks, AWP_NCHUNKS);
    if (train) return p.geo_rows ? launch_awp_embed_t<PREC, true, true>(p, st) : launch_awp_embed_t<PREC, true, false>(p, st);
    return p.geo_rows ? launch_awp_embed_t<PREC, false, true>(p, st) : launch_awp_embed_t<PREC, false, false>(p, st);
}

// ---- backward ----------------------------------------------------------------------------------------------------------------
// d h_local rows [n, 64] x loss scale -> the gradient fragments of the last layer, multiplied by its ReLU pattern
template <int PREC>
__global__ __launch_bounds__(256) void k_awp_rows_to_frags(const float* __restrict__ rows, long nsamp, long tiles, const unsigned* __restrict__ maxbits,
                                                           char* __restrict__ store) {
    typedef POps<PREC> O;
    pipe_fp16_saturate<PREC>();
    constexpr int KW = AWP_W / 16, last = awpstore::E0 + (AWP_D - 1) * KW, dlast = awpstore::D_E0 + (AWP_D - 1) * KW;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x, tile = idx / (64 * KW);
    if (tile >= tiles) return;
    const int j = (int)((idx / 64) % KW), lane = idx & 63, n = lane & 31, h = lane >> 5;
    const long smp = tile * 32 + n;
    char* a = store + tile * awpstore::TILE_BYTES + lane * 16;
    const W4 act = frag_load<W4>(a, last + j);
    const float s = grad_scale(*maxbits, false);
    f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
    if (smp < nsamp) {
        const float* r = rows + smp * AWP_W + 16 * j + 4 * h;
        lo = *reinterpret_cast<const f32x4*>(r);
        hi = *reinterpret_cast<const f32x4*>(r + 8);
    }
    typename O::B b;
    O::template set_pair<false>(b, 0, lo[0] * s, lo[1] * s);
    O::template set_pair<false>(b, 1, lo[2] * s, lo[3] * s);
    O::template set_pair<false>(b, 2, hi[0] * s, hi[1] * s);
    O::template set_pair<false>(b, 3, hi[2] * s, hi[3] * s);
#pragma unroll
    for (int e = 0; e < 4; ++e) b.w[e] = mask_word(b.w[e], act.w[e]);
    act_store(a, dlast + j, b);
}

// max |v| of nfrag gradient fragments per tile, in true units (loss scale removed) -> out (float bits).  Grid-stride over the
// (tile, fragment, lane) items, one atomicMax per block (one per wavefront on a single word cost 3.7 ms at 1.3 M samples).
template <int PREC>
__global__ __launch_bounds__(256) void k_frag_absmax(const char* __restrict__ store, long tile_bytes, int slot, int nfrag, long tiles,
                                                     const unsigned* __restrict__ maxbits, unsigned* __restrict__ out) {
    __shared__ float part[4];
    const long total = tiles * 64 * nfrag;
    float m = 0.f;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long tile = idx / (64 * nfrag);
        const int j = (int)((idx / 64) % nfrag), lane = idx & 63;
        const W4 f = frag_load<W4>(store + tile * tile_bytes + lane * 16, slot + j);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const unsigned short bits = (unsigned short)(f.w[e >> 1] >> (16 * (e & 1)));
            float v;
            if constexpr (PREC == EVD_PREC_BF16) v = __uint_as_float((unsigned)bits << 16);
            else v = (float)__builtin_bit_cast(_Float16, bits);
            m = fmaxf(m, fabsf(v));
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
        if (m > 0.f) atomicMax(out, __float_as_uint(m * grad_scale(*maxbits, true)));
    }
}

template <int PREC> static int run_awp_backward(const AwpBwdPlan& b, hipStream_t st) {
    using namespace awpstore;
    constexpr int KW = AWP_W / 16, T = AWP_W / 32;
    int rc;
    unsigned* words = reinterpret_cast<unsigned*>(b.store + b.tiles * TILE_BYTES);       // trailer: loss-scale word, max |d geo|
    EVD_HIP(hipMemsetAsync(words, 0, 2 * sizeof(unsigned), st));
    if (b.d_h_absmax) {
        EVD_HIP(hipMemcpyAsync(words, b.d_h_absmax, sizeof(unsigned), hipMemcpyDeviceToDevice, st));
    } else {
        hipLaunchKernelGGL(k_absmax, dim3(512), dim3(256), 0, st, b.d_h_local, b.nsamp * AWP_W, words);
        EVD_LAUNCH_CHECK();
    }
    // round 5: the whole chain in one launch with the tile's gradient resident in registers (awp_bwd_fused.h); EVD_AWP_BWD_FUSE=0: the
    // per-layer chain below (A/B, and what the fused kernel is tested against)
    static const bool fuse = [] { const char* e = getenv("EVD_AWP_BWD_FUSE"); return !(e && e[0] == '0'); }();
    if (fuse) {
        const int blocks = (int)(cdiv(b.tiles, 4L) < b.wgrad_blocks ? cdiv(b.tiles, 4L) : b.wgrad_blocks);
        AwpBwdFusedParams fp;
        fp.d_h_local = b.d_h_local; fp.nsamp = b.nsamp; fp.tiles = b.tiles; fp.store = b.store; fp.words = words; fp.partial = b.partial;
        for (int l = 0; l < AWP_D; ++l) fp.wt[l] = b.wt[l];
        EVD_SET_MAX_LDS((&k_awp_bwd_fused<PREC>), (size_t)awpf::LDS_BYTES);
        hipLaunchKernelGGL((k_awp_bwd_fused<PREC>), dim3((unsigned)blocks), dim3(256), (size_t)awpf::LDS_BYTES, st, fp);
        EVD_LAUNCH_CHECK();
        const AwpBwdGrads& g = b.grads;
        WreduceJobs jobs;
        for (int i = 0; i < WREDUCE_MAX_JOBS; ++i) {
            const int l = i < 3 ? 3 - i : 1;            // jobs 0, 1, 2: layers 3, 2, 1; the rest empty
            WreduceParams& q = jobs.j[i];
            q.partial = b.partial + (long)awpf::acc0(l) * 1024; q.nparts = blocks; q.RT = (i < 3 && g.w[l]) ? 2 : 0; q.CT = 2; q.NC = 2;
            q.rowmap = b.maps + AMAP_H; q.colmap = b.maps + AMAP_H; q.dW = g.w[l]; q.ld = AWP_W; q.db = nullptr;
            q.maxbits = words; q.accum = 0; q.part_stride = (long)awpf::NBLK * 1024;
        }
        hipLaunchKernelGGL(k_wgrad_reduce_jobs, dim3(2 * 2 * 4, WREDUCE_MAX_JOBS), dim3(256), 0, st, jobs);
        EVD_LAUNCH_CHECK();
        BiasColsParams bp;
        bp.partial = b.partial + (long)awpf::A_BIAS * 1024; bp.nparts = blocks; bp.part_stride = (long)awpf::NBLK * 1024; bp.ncols = 6;
        for (int l = 1; l < AWP_D; ++l)
            for (int yb = 0; yb < 2; ++yb) { bp.rowmap[2 * (3 - l) + yb] = b.maps + AMAP_H + 32 * yb; bp.db[2 * (3 - l) + yb] = g.b[l]; }
        bp.maxbits = words; bp.accum = 0;
        hipLaunchKernelGGL(k_bias_cols_reduce, dim3(6 * 32 / 4), dim3(256), 0, st, bp);
        EVD_LAUNCH_CHECK();
        // layer 0's weight gradient from the stored (d e0, geo) fragments: the per-layer kernel (behind the reduces above: they share `partial`)
        {
            if (g.w[0]) {
                const int wb = (int)(b.tiles < b.wgrad_blocks ? b.tiles : b.wgrad_blocks);
                WgradParams wp;
                wp.store = b.store; wp.tiles = b.tiles; wp.tile_bytes = TILE_BYTES; wp.y_slot = D_E0; wp.x_slot = GEO; wp.bias = 1; wp.partial = b.partial;
                int r = launch_wgrad<PREC, T, AWP_IN / 32, false>(wp, wb, st);
                if (r) return r;
                WreduceParams q;
                q.partial = b.partial; q.nparts = wb; q.RT = T; q.CT = AWP_IN / 32; q.NC = q.CT + 1;
                q.rowmap = b.maps + AMAP_H; q.colmap = b.maps + AMAP_GEO; q.dW = g.w[0]; q.ld = AWP_IN; q.db = g.b[0]; q.maxbits = words;
                hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((long)T * q.NC * 4)), dim3(256), 0, st, q);
                EVD_LAUNCH_CHECK();
            }
        }
        if (b.d_geo_rows) {
            hipLaunchKernelGGL((k_frags_to_rows<PREC>), dim3((unsigned)cdiv(b.tiles * 64 * (AWP_IN / 16), 256L)), dim3(256), 0, st, (const char*)b.store, TILE_BYTES,
                               D_GEO, AWP_IN / 16, b.nsamp, words, b.d_geo_rows, AWP_IN);
            EVD_LAUNCH_CHECK();
        }
        return EVD_OK;
    }
    hipLaunchKernelGGL((k_awp_rows_to_frags<PREC>), dim3((unsigned)cdiv(b.tiles * 64 * KW, 256L)), dim3(256), 0, st, b.d_h_local, b.nsamp, b.tiles, words, b.store);
    EVD_LAUNCH_CHECK();
    auto dgrad = [&](int l, int in_slot, int mask_slot, int out_slot) {
        DgradParams p;
        p.wstream = b.wt[l]; p.store = b.store; p.tile_bytes = TILE_BYTES; p.in_slot = in_slot; p.extra_slot = -1; p.mask_slot = mask_slot; p.out_slot = out_slot;
        return p;
    };
    auto wgrad = [&](auto launch, int RT, int CT, int y_slot, int x_slot, int xmap, float* dW, int ld, float* db) -> int {
        if (!dW) return EVD_OK;
        const int blocks = (int)(b.tiles < b.wgrad_blocks ? b.tiles : b.wgrad_blocks);
        WgradParams p;
        p.store = b.store; p.tiles = b.tiles; p.tile_bytes = TILE_BYTES; p.y_slot = y_slot; p.x_slot = x_slot; p.bias = 1; p.partial = b.partial;
        int r = launch(p, blocks, st);
        if (r) return r;
        WreduceParams q;
        q.partial = b.partial; q.nparts = blocks; q.RT = RT; q.CT = CT; q.NC = CT + 1;
        q.rowmap = b.maps + AMAP_H; q.colmap = b.maps + xmap; q.dW = dW; q.ld = ld; q.db = db; q.maxbits = words;
        hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((long)RT * q.NC * 4)), dim3(256), 0, st, q);
        EVD_LAUNCH_CHECK();
        return EVD_OK;
    };
    const AwpBwdGrads& g = b.grads;
    for (int l = AWP_D - 1; l >= 1; --l) {
        if ((rc = wgrad(launch_wgrad<PREC, T, T, false>, T, T, D_E0 + KW * l, E0 + KW * (l - 1), AMAP_H, g.w[l], AWP_W, g.b[l]))) return rc;
        // d e_{l-1} = (W_l^T d e_l) . [e_{l-1} > 0]: the ReLU pattern from the saved activation fragments
        if ((rc = launch_dgrad<PREC, KW, T, KW, false, 1>(dgrad(l, D_E0 + KW * l, E0 + KW * (l - 1), D_E0 + KW * (l - 1)), b.tiles, st))) return rc;
    }
    if ((rc = wgrad(launch_wgrad<PREC, T, AWP_IN / 32, false>, T, AWP_IN / 32, D_E0, GEO, AMAP_GEO, g.w[0], AWP_IN, g.b[0]))) return rc;
    {   // d geo (no activation); its maximum (true units, for the fine level's rescaling) is taken by the kernel that forms it
        DgradParams dp = dgrad(0, D_E0, -1, D_GEO);
        dp.absmax_out = words + 1;
        dp.maxbits = words;
        if ((rc = launch_dgrad<PREC, KW, AWP_IN / 32, KW, false, 0>(dp, b.tiles, st))) return rc;
    }
    if (b.d_geo_rows) {
        hipLaunchKernelGGL((k_frags_to_rows<PREC>), dim3((unsigned)cdiv(b.tiles * 64 * (AWP_IN / 16), 256L)), dim3(256), 0, st, (const char*)b.store, TILE_BYTES,
                           D_GEO, AWP_IN / 16, b.nsamp, words, b.d_geo_rows, AWP_IN);
        EVD_LAUNCH_CHECK();
    }
    return EVD_OK;
}

}  // namespace evd
