// Software-pipelined PDRF level network (fine: hidden 256, geo 128, 64 feature channels in; the training path also builds the
// coarse 64 / 15 / 32 level on it; reference
// networks/pdrf/voxnerf.py:210-221,240-254 with the blurfactory dimensions): sigma net 127 -> 256 -> 1 + 128,
// colour net 155 -> 256 -> 256 -> 3 (sigmoid), on the machinery of mlp_pipe.h.  One straight-line stream of 368
// MFMAs per wavefront; layer table below.  The coarse level (64-wide) runs the same table: its training forward on k_voxel_mlp_pipe (TRAIN), its
// render pass -- and its training forward in the split-float16 modes -- on k_voxel_mlp_resident (round 6: the whole weight stream resident in LDS,
// persistent workgroups, no barrier); with per-sample feature rows wanted it takes kernel_voxel.hip's generic kernel.
#pragma once

#include "mlp_pipe.h"
#include "voxel.h"
#ifndef EVD_RES_NT
#define EVD_RES_NT 512      // threads of k_voxel_mlp_resident: two wavefronts per SIMD (198 registers in the split-float16 mode); -DEVD_RES_NT=256: one (A/B)
#endif

namespace evd {

// Fragment slots of the PDRF activation / gradient store (training path), per 32-sample tile, for a level with hidden width HD,
// G geo channels and FT feature channels in: what the forward saves, then the gradients the dgrad chain hands to wgrad.
template <int HD, int G, int FT> struct VStore {
    static constexpr int KS = HD / 16, KF = FT / 16, GT = (G + 31) / 32;
    // (the direction encoding sits behind the geo fragments: color_net.0's input [geo | PE(dirs)] is one run of fragments for its wgrad)
    static constexpr int IN0 = 0, HID = IN0 + KF + PE_KS, GEO = HID + KS, DIRPE = GEO + 2 * GT, C0 = DIRPE + PEV_KS, C1 = C0 + KS, FWD_END = C1 + KS;
    // gradient slots; D_GEO is followed by the 2 direction-encoding gradient fragments, D_FTS by the 4 point-encoding ones (both
    // produced by the same dgrad layer as their neighbours, in the encodings' own fragment arrangement)
    // (M_C1 sits right behind G_COL: the fused backward of color_net.1 fetches [d colour fragment | c1's ReLU bits] as ONE pair of
    // DMA pieces and forms d c1 from them in registers, nerf_train_kernel.h k_wgrad_dgrad YGEN)
    static constexpr int G_COL = FWD_END, M_C1 = G_COL + 1, G_SIG = M_C1 + 1, D_C1 = G_SIG + 1, D_C0 = D_C1 + KS, D_GEO = D_C0 + KS, D_DIRPE = D_GEO + 2 * GT,
                         D_HID = D_DIRPE + PEV_KS, D_FTS = D_HID + KS, D_PE = D_FTS + 2 * ((FT + 31) / 32),
                         M_HID = D_PE + PE_KS, M_C0 = M_HID + 1,      // ReLU patterns as bit masks (mlp_pipe.h frag_bits)
                         TILE_FRAGS = M_C0 + 1;
    static constexpr long TILE_BYTES = (long)TILE_FRAGS * 1024;          // the single-product half-precision modes
    static constexpr long tile_bytes(int prec) { return (long)TILE_FRAGS * frag_bytes(prec); }   // split-float16: (hi, lo) pairs in 2 KiB slots
};

// layer table of one PDRF level: HD hidden width, G geo channels (15: one zero-padded tile; 128: four), FT feature channels in
template <class C, int HD_, int G_, int FT_, bool FEAT, bool TRAIN = false> struct VoxNet {
    static constexpr int HD = HD_, G = G_, FT = FT_;
    typedef VStore<HD, G, FT> VS;
    static constexpr int T = HD / 32, KS = HD / 16, KF = FT / 16, GT = (G + 31) / 32, GK = 2 * GT, FPC = C::FPC, PD = C::PD;
    static constexpr int slot(int s) { return TRAIN ? s : -1; }
    // sigma_net.0 on cat([fts, PE(pts)]) (voxnerf.py:214): k-steps [fts_0.. | pe_0..3]
    typedef LayerDesc<KF + PE_KS, T, 1, true, false, 0, 0, false, 0, 0, 0, false, 0, -1, false, 1, slot(VS::HID), -1, slot(VS::M_HID)> L0;
    static constexpr int F1 = T * (KF + PE_KS);
    // sigma_net.1 row 0 = sigma (float32 out) ...
    typedef LayerDesc<KS, 1, 1, false, true, 0, F1, false, F1 % PD, L0::PAR_OUT, 1, true, KS - 2, -1, false, 1, -1, slot(VS::HID + KS - 2), -1, slot(VS::M_HID),
                      2 * (T - 1)> Sigma;
    static constexpr int F2 = F1 + KS;
    // ... rows 1..G = geo features (no activation; the per-sample feature AWP consumes, voxnerf.py:221)
    typedef LayerDesc<KS, GT, 1, false, false, 0, F2, false, F2 % PD, Sigma::PAR_OUT, 0, false, 0, -1, FEAT, 1, slot(VS::GEO)> Geo;
    static constexpr int F3 = F2 + GT * KS;
    // color_net.0 on cat([geo, PE(dirs)]) (voxnerf.py:248): k-steps [geo_0.. | dir_0..1]; geo's last tile lands at GK - 2, GK - 1
    typedef LayerDesc<GK + PEV_KS, T, 1, true, false, 0, F3, false, F3 % PD, Geo::PAR_OUT, 1, false, GK - 2, FEAT ? GT - 1 : -1, false, 1, slot(VS::C0),
                      slot(VS::GEO + GK - 2), slot(VS::M_C0)> C0;
    static constexpr int F4 = F3 + T * (GK + PEV_KS);
    typedef LayerDesc<KS, T, 1, true, false, 0, F4, false, F4 % PD, C0::PAR_OUT, 1, true, KS - 2, -1, false, 1, slot(VS::C1), slot(VS::C0 + KS - 2), slot(VS::M_C1),
                      slot(VS::M_C0), 2 * (T - 1)> C1;
    static constexpr int F5 = F4 + T * KS;
    typedef LayerDesc<KS, 1, 1, false, true, 0, F5, true, F5 % PD, C1::PAR_OUT, 1, true, KS - 2, -1, false, 0, -1, slot(VS::C1 + KS - 2), -1, slot(VS::M_C1),
                      2 * (T - 1)> C2;
    static constexpr int NCH = cceil(F5 + KS, FPC);
    // LDS bias image in stream order: the sigma net has no biases (zeros)
    static constexpr int B_SIG = T * 32, B_GEO = B_SIG + 32, B_C0 = B_GEO + GT * 32, B_C1 = B_C0 + T * 32, B_C2 = B_C1 + T * 32, B_END = B_C2 + 32;
    static_assert(F1 % PD == 0 && F3 % PD == 0 && F4 % PD == 0, "prefetch ring phase");
    static_assert(!FEAT || G % 32 == 0, "float32 feature rows are written 32 at a time");
};
template <class C, bool FEAT> using VoxFineNet = VoxNet<C, 256, 128, 64, FEAT, false>;

template <int PREC, int HD, int G, int FT, int NS, int NT, bool FEAT, int CB, int OCC, bool TRAIN, bool HI_ONLY = false>
__global__ __launch_bounds__(NT, OCC) void k_voxel_mlp_pipe(const VoxMlpParams p) {
    typedef PipeCfg<PREC, NS, NT, CB, HI_ONLY> C;
    typedef typename C::O O;
    typedef typename O::B B;
    typedef VoxNet<C, HD, G, FT, FEAT, TRAIN> N;
    typedef typename N::VS VS;
    constexpr int T = N::T, KS = N::KS, KF = N::KF, GK = N::GK;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef PStream<C, FEAT || TRAIN, N::NCH> ST;

    pipe_fp16_saturate<PREC>();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, h = lane >> 5;
    ST st;
    st.start_issue(p.wstream, smem, tid);
    float* bias = reinterpret_cast<float*>(smem + C::RING);
    // bias image: zeros for the sigma net, then the colour-net biases (p.bias = 512 zeros + colour biases, evd_voxel_api.hip)
    for (int i = tid; i < N::B_END; i += NT) bias[i] = i < N::B_C0 ? 0.f : p.bias[512 + (i - N::B_C0)];
    B* stash = reinterpret_cast<B*>(smem + C::RING + C::BIAS_FLOATS * 4 + wave * C::STASH_PER_WAVE) + lane;

    long sidx[NS];
    bool valid[NS];
    char* actl[NS];
    B in0[NS][KF + PE_KS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const long smp = (long)blockIdx.x * C::SAMPLES + wave * (NS * 32) + s * 32 + n;
        valid[s] = smp < p.nsamp;
        sidx[s] = valid[s] ? smp : p.nsamp - 1;
        actl[s] = TRAIN ? p.act + (smp >> 5) * VS::tile_bytes(C::STORE_PREC) + lane * 16 : nullptr;
        float pts[3], vd[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            pts[c] = p.pts[sidx[s] * 3 + c];
            vd[c] = p.viewdirs[(sidx[s] / p.S) * p.vd_stride + c];
        }
        // feature k-steps in natural order: B position 8h + e of k-step j <-> feature 16 j + 8 h + e
        const float* f = p.fts + sidx[s] * (long)p.ft_stride + 8 * h;
#pragma unroll
        for (int j = 0; j < KF; ++j) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(f + 16 * j), b = *reinterpret_cast<const f32x4*>(f + 16 * j + 4);
            O::template set_pair<false>(in0[s][j], 0, a[0], a[1]);
            O::template set_pair<false>(in0[s][j], 1, a[2], a[3]);
            O::template set_pair<false>(in0[s][j], 2, b[0], b[1]);
            O::template set_pair<false>(in0[s][j], 3, b[2], b[3]);
        }
        B pe[PE_KS], pev[PEV_KS];
        encode_pairs<C, PE_L, PE_KS>(pts, h, pe);
        encode_pairs<C, PE_LV, PEV_KS>(vd, h, pev);
#pragma unroll
        for (int j = 0; j < PE_KS; ++j) in0[s][KF + j] = pe[j];
#pragma unroll
        for (int j = 0; j < PEV_KS; ++j) stash[(s * C::STASH_FRAGS + j) * 64] = pev[j];   // parked until the colour net
        if constexpr (TRAIN) {
#pragma unroll
            for (int j = 0; j < KF + PE_KS; ++j) pipe_act_store<C>(actl[s], VS::IN0 + j, in0[s][j]);
#pragma unroll
            for (int j = 0; j < PEV_KS; ++j) pipe_act_store<C>(actl[s], VS::DIRPE + j, pev[j]);
        }
    }
    float* frow[NS];
    float* nofrow[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        frow[s] = (FEAT && p.feature && valid[s]) ? p.feature + sidx[s] * N::G : nullptr;
        nofrow[s] = nullptr;
    }
    st.start_wait();
    Pipe<C> pp;
    pipe_prime<C, typename N::L0>(st, pp, bias, lane);

    B hid[NS][KS], none[NS][1];
    pipe_layer<C, typename N::L0, ST, KS, TRAIN>(st, pp, in0, hid, nullptr, bias, lane, nofrow, actl);
    float sig[NS][4], col[NS][4];
    pipe_layer<C, typename N::Sigma, ST, 1, TRAIN>(st, pp, hid, none, sig, bias + N::B_SIG, lane, nofrow, actl);
    B cin[NS][GK + PEV_KS];
    pipe_layer<C, typename N::Geo, ST, GK + PEV_KS, TRAIN>(st, pp, hid, cin, nullptr, bias + N::B_GEO, lane, frow, actl);
    {
        const B* sp = stash;
        asm volatile("" : "+v"(sp));
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int j = 0; j < PEV_KS; ++j) cin[s][GK + j] = sp[(s * C::STASH_FRAGS + j) * 64];
    }
    B c0[NS][KS], c1[NS][KS];
    pipe_layer<C, typename N::C0, ST, KS, TRAIN>(st, pp, cin, c0, nullptr, bias + N::B_C0, lane, frow, actl);
    pipe_layer<C, typename N::C1, ST, KS, TRAIN>(st, pp, c0, c1, nullptr, bias + N::B_C1, lane, nofrow, actl);
    pipe_layer<C, typename N::C2, ST, 1, TRAIN>(st, pp, c1, none, col, bias + N::B_C2, lane, nofrow, actl);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (h == 0 && valid[s]) {
            f32x4 o;
            o[0] = sig[s][0];
#pragma unroll
            for (int c = 0; c < 3; ++c) o[1 + c] = 1.f / (1.f + expf(-col[s][c]));      // torch.sigmoid(h) voxnerf.py:252
            *reinterpret_cast<f32x4*>(p.raw + sidx[s] * 4) = o;
        }
    }
}

// (Measured and dropped: 256-thread workgroups walking the same stream in 8 KiB chunks -- 72 KiB of LDS, two workgroups per CU
// overlapping each other's prologue and barrier waits -- run 1 % slower: the overlap is paid back by twice the L2 -> LDS weight
// traffic and DMA issue.  PipeCfg keeps the chunk size as a parameter.)
template <int PREC, bool FEAT>
static int launch_voxel_pipe(const VoxMlpParams& p, hipStream_t st) {
    constexpr bool half = is_half_prec(PREC);
    constexpr int NT = half ? 512 : 256;      // the split-float16 B fragments need the whole register file: one wavefront per SIMD
    constexpr int OCC = half ? 2 : 1;
    typedef PipeCfg<PREC, 1, NT> C;
    typedef VoxFineNet<C, FEAT> N;
    const long blocks = cdiv(p.nsamp, C::SAMPLES);
    const size_t lds = C::TOTAL;
    EVD_SET_MAX_LDS((&k_voxel_mlp_pipe<PREC, 256, 128, 64, 1, NT, FEAT, PIPE_CB, OCC, false>), lds);
    if (p.nchunks != N::NCH) return fail(EVD_E_INVALID, "evd_voxel: packed stream has %d chunks, kernel expects %d", p.nchunks, N::NCH);
    hipLaunchKernelGGL((k_voxel_mlp_pipe<PREC, 256, 128, 64, 1, NT, FEAT, PIPE_CB, OCC, false>), dim3((unsigned)blocks), dim3(NT), lds, st, p);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

// training variant (keeps the activations), either level; the stream is the level's pipe stream (evd_voxel_api.hip).  The store is
// tiled in groups of 8 tiles (256 samples) and the backward walks all of them: the grid covers the padding tiles too.
// HI_ONLY (PREC = EVD_PREC_F16X3): the store is the single-product float16 mode's (mlp_pipe.h PipeCfg) -- the coarse level of a
// training forward in EVD_PREC_F16C
template <int PREC, int HD, int G, int FT, bool HI_ONLY> static int launch_voxel_resident_train(const VoxMlpParams& p, hipStream_t st);
template <int PREC, int HD, int G, int FT, bool FEAT = false, bool HI_ONLY = false>
static int launch_voxel_train_fwd(const VoxMlpParams& p, hipStream_t st) {
    // the 64-wide level in the split-float16 arithmetic (the coarse level of the f16c / f16m / f16x3 training modes): weight stream resident in
    // LDS, persistent workgroups -- iteration f16c 11.37 -> 11.24 ms, f16m 12.93 -> 12.81; in the single-product modes this kernel already runs two
    // workgroups per CU and the resident form measures 197 against 189 us: not used there (profiles/r06_coarse_train_ab.log).  EVD_COARSE_FORM=pipe: this kernel
    if constexpr (HD == 64 && !FEAT && PREC == EVD_PREC_F16X3) {
        static const bool pipe_form = [] { const char* e = getenv("EVD_COARSE_FORM"); return e && !strcmp(e, "pipe"); }();
        if (!pipe_form && p.nsamp >= 65536) return launch_voxel_resident_train<PREC, HD, G, FT, HI_ONLY>(p, st);
    }
    constexpr int NT = is_half_prec(PREC) ? 512 : 256, OCC = is_half_prec(PREC) ? 2 : 1;     // split-float16: one wavefront per SIMD
    typedef PipeCfg<PREC, 1, NT, PIPE_CB, HI_ONLY> C;
    typedef VoxNet<C, HD, G, FT, FEAT, true> N;
    const long blocks = cdiv(p.nsamp, 256L) * (256 / C::SAMPLES);
    const size_t lds = C::TOTAL;
    EVD_SET_MAX_LDS((&k_voxel_mlp_pipe<PREC, HD, G, FT, 1, NT, FEAT, PIPE_CB, OCC, true, HI_ONLY>), lds);
    if (p.nchunks != N::NCH) return fail(EVD_E_INVALID, "evd_voxel: packed stream has %d chunks, kernel expects %d", p.nchunks, N::NCH);
    if (!p.act) return fail(EVD_E_INVALID, "evd_voxel: training launch without an activation store");
    hipLaunchKernelGGL((k_voxel_mlp_pipe<PREC, HD, G, FT, 1, NT, FEAT, PIPE_CB, OCC, true, HI_ONLY>), dim3((unsigned)blocks), dim3(NT), lds, st, p);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

// inference on the software pipeline for ANY level, on the stream the level's training forward uses (same layer table, no activation store):
// the 64-wide coarse level's render pass (the generic kernel took 86 us per 4096 x 64 samples in f16x3 -- 11 % of a c2f render in the
// compensated mode, whose coarse level runs float32-grade)
template <int PREC, int HD, int G, int FT>
static int launch_voxel_pipe_level(const VoxMlpParams& p, hipStream_t st) {
    // (split-float16: one wavefront per SIMD as in the fine level -- with 512 threads the ring + stash of PipeCfg exceed the 160 KiB of LDS;
    // measured 86 us per 4096 x 64 samples either way, the same as the generic kernel: the 64-wide level is prologue-bound, not MFMA-bound)
    constexpr int NT = is_half_prec(PREC) ? 512 : 256, OCC = is_half_prec(PREC) ? 2 : 1;
    typedef PipeCfg<PREC, 1, NT> C;
    typedef VoxNet<C, HD, G, FT, false, false> N;
    const long blocks = cdiv(p.nsamp, C::SAMPLES);
    const size_t lds = C::TOTAL;
    EVD_SET_MAX_LDS((&k_voxel_mlp_pipe<PREC, HD, G, FT, 1, NT, false, PIPE_CB, OCC, false>), lds);
    if (p.nchunks != N::NCH) return fail(EVD_E_INVALID, "evd_voxel: packed stream has %d chunks, kernel expects %d", p.nchunks, N::NCH);
    if (p.feature || p.act) return fail(EVD_E_INVALID, "evd_voxel: the level's pipelined inference pass writes raw only");
    hipLaunchKernelGGL((k_voxel_mlp_pipe<PREC, HD, G, FT, 1, NT, false, PIPE_CB, OCC, false>), dim3((unsigned)blocks), dim3(NT), lds, st, p);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}
// Round 6: the same layer table, arithmetic and order of operations (results bit for bit those of k_voxel_mlp_pipe) for a level whose whole
// weight stream fits LDS -- the 64-wide coarse level of the shipped c2f configs.  k_voxel_mlp_pipe is built for 256-wide layers: every
// workgroup streams the weights through a 4-slot ring with a counted wait and a barrier per 16 KiB chunk and lives for ONE group of sample
// tiles, so this level's render pass was all prologue and hand-over (86-90 us per 4096 x 64 samples, 14 % of a c2f render, whatever the
// arithmetic mode).  Here the workgroups are persistent (one per CU), copy the stream ONCE (mlp_pipe.h PResident) and walk the sample tiles
// without a barrier; the direction encoding stays in registers (no stash).
// TRAIN / HI_ONLY: the training forward of the level (k_voxel_mlp_pipe's TRAIN variant: every completed block also goes to the activation store).
template <int PREC, int HD, int G, int FT, int NT, bool TRAIN = false, bool HI_ONLY = false, bool REV = false>
__global__ __launch_bounds__(NT, 1) void k_voxel_mlp_resident(const VoxMlpParams p) {
    typedef PipeCfg<PREC, 1, NT, PIPE_CB, HI_ONLY> C;
    typedef typename C::O O;
    typedef typename O::B B;
    typedef VoxNet<C, HD, G, FT, false, TRAIN> N;
    typedef typename N::VS VS;
    constexpr int KS = N::KS, KF = N::KF, GK = N::GK;
    typedef PResident<C, N::NCH> ST;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    pipe_fp16_saturate<PREC>();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, h = lane >> 5;
    ST st;
    st.start_issue(p.wstream, smem, tid);
    float* bias = reinterpret_cast<float*>(smem + ST::BYTES);
    for (int i = tid; i < N::B_END; i += NT) bias[i] = i < N::B_C0 ? 0.f : p.bias[512 + (i - N::B_C0)];
    st.start_wait();
    // (TRAIN: the store is tiled in groups of 8 sample tiles and the backward walks all of them -- the padding tiles are written too)
    const long ngroups = TRAIN ? (p.nsamp + 255) / 256 * (256 / C::SAMPLES) : (p.nsamp + C::SAMPLES - 1) / C::SAMPLES;
    for (long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const long smp = grp * C::SAMPLES + wave * 32 + n;
        const bool valid = smp < p.nsamp;
        const long sidx = valid ? smp : p.nsamp - 1;
        char* actl[1] = {TRAIN ? p.act + (smp >> 5) * VS::tile_bytes(C::STORE_PREC) + lane * 16 : nullptr};
        B in0[1][KF + PE_KS], pev[PEV_KS];
        {
            float pts[3], vd[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                pts[c] = p.pts[sidx * 3 + c];
                vd[c] = p.viewdirs[(sidx / p.S) * p.vd_stride + c];
            }
            const float* f = p.fts + sidx * (long)p.ft_stride + 8 * h;
#pragma unroll
            for (int j = 0; j < KF; ++j) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(f + 16 * j), b = *reinterpret_cast<const f32x4*>(f + 16 * j + 4);
                O::template set_pair<false>(in0[0][j], 0, a[0], a[1]);
                O::template set_pair<false>(in0[0][j], 1, a[2], a[3]);
                O::template set_pair<false>(in0[0][j], 2, b[0], b[1]);
                O::template set_pair<false>(in0[0][j], 3, b[2], b[3]);
            }
            B pe[PE_KS];
            if constexpr (REV) {
                encode_pairs_rev<C, PE_L, PE_KS>(pts, h, pe);
                encode_pairs_rev<C, PE_LV, PEV_KS>(vd, h, pev);
            } else {
                encode_pairs<C, PE_L, PE_KS>(pts, h, pe);
                encode_pairs<C, PE_LV, PEV_KS>(vd, h, pev);
            }
#pragma unroll
            for (int j = 0; j < PE_KS; ++j) in0[0][KF + j] = pe[j];
            if constexpr (TRAIN) {
#pragma unroll
                for (int j = 0; j < KF + PE_KS; ++j) pipe_act_store<C>(actl[0], VS::IN0 + j, in0[0][j]);
#pragma unroll
                for (int j = 0; j < PEV_KS; ++j) pipe_act_store<C>(actl[0], VS::DIRPE + j, pev[j]);
            }
        }
        float* nofrow[1] = {nullptr};
        Pipe<C> pp;
        pipe_prime<C, typename N::L0>(st, pp, bias, lane);
        B hid[1][KS], none[1][1];
        pipe_layer<C, typename N::L0, ST, KS, TRAIN>(st, pp, in0, hid, nullptr, bias, lane, nofrow, actl);
        float sig[1][4], col[1][4];
        pipe_layer<C, typename N::Sigma, ST, 1, TRAIN>(st, pp, hid, none, sig, bias + N::B_SIG, lane, nofrow, actl);
        B cin[1][GK + PEV_KS];
        pipe_layer<C, typename N::Geo, ST, GK + PEV_KS, TRAIN>(st, pp, hid, cin, nullptr, bias + N::B_GEO, lane, nofrow, actl);
#pragma unroll
        for (int j = 0; j < PEV_KS; ++j) cin[0][GK + j] = pev[j];
        B c0[1][KS], c1[1][KS];
        pipe_layer<C, typename N::C0, ST, KS, TRAIN>(st, pp, cin, c0, nullptr, bias + N::B_C0, lane, nofrow, actl);
        pipe_layer<C, typename N::C1, ST, KS, TRAIN>(st, pp, c0, c1, nullptr, bias + N::B_C1, lane, nofrow, actl);
        pipe_layer<C, typename N::C2, ST, 1, TRAIN>(st, pp, c1, none, col, bias + N::B_C2, lane, nofrow, actl);
        if (h == 0 && valid) {
            f32x4 o;
            o[0] = sig[0][0];
#pragma unroll
            for (int c = 0; c < 3; ++c) o[1 + c] = 1.f / (1.f + expf(-col[0][c]));      // torch.sigmoid(h) voxnerf.py:252
            *reinterpret_cast<f32x4*>(p.raw + sidx * 4) = o;
        }
    }
}

// the level's TRAINING forward on the resident stream (HD = 64 levels; the caller falls back to k_voxel_mlp_pipe's TRAIN variant otherwise)
template <int PREC, int HD, int G, int FT, bool HI_ONLY>
static int launch_voxel_resident_train(const VoxMlpParams& p, hipStream_t st) {
    constexpr int NT = EVD_RES_NT;
    typedef PipeCfg<PREC, 1, NT, PIPE_CB, HI_ONLY> C;
    typedef VoxNet<C, HD, G, FT, false, true> N;
    typedef PResident<C, N::NCH> ST;
    constexpr size_t lds = (size_t)ST::BYTES + (size_t)C::BIAS_FLOATS * 4;
    static_assert(lds <= 160 * 1024, "the level's stream is resident in LDS");
    if (p.nchunks != N::NCH) return fail(EVD_E_INVALID, "evd_voxel: packed stream has %d chunks, kernel expects %d", p.nchunks, N::NCH);
    if (!p.act) return fail(EVD_E_INVALID, "evd_voxel: training launch without an activation store");
    int cus = 256;
    { int dev = 0, v = 0; if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v; }
    const long groups = cdiv(p.nsamp, 256L) * (256 / C::SAMPLES);
    EVD_SET_MAX_LDS((&k_voxel_mlp_resident<PREC, HD, G, FT, NT, true, HI_ONLY>), lds);
    hipLaunchKernelGGL((k_voxel_mlp_resident<PREC, HD, G, FT, NT, true, HI_ONLY>), dim3((unsigned)(groups < cus ? groups : cus)), dim3(NT), lds, st, p);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

template <int PREC, int HD, int G, int FT>
static int launch_voxel_resident_level(const VoxMlpParams& p, hipStream_t st) {
    // EVD_COARSE_FORM=pipe (developer switch): the streaming kernel of rounds 3-5 (A/B)
    static const bool pipe_form = [] { const char* e = getenv("EVD_COARSE_FORM"); return e && !strcmp(e, "pipe"); }();
    // (a launch of less than one tile group per CU and wavefront slot does not repay the 48-80 KiB copy: 333 x 17 samples 17.2 against 14.4 us)
    if (pipe_form || p.nsamp < 65536) return launch_voxel_pipe_level<PREC, HD, G, FT>(p, st);
    constexpr int NT = EVD_RES_NT;
    typedef PipeCfg<PREC, 1, NT> C;
    typedef VoxNet<C, HD, G, FT, false, false> N;
    typedef PResident<C, N::NCH> ST;
    constexpr size_t lds = (size_t)ST::BYTES + (size_t)C::BIAS_FLOATS * 4;
    static_assert(lds <= 160 * 1024, "the level's stream is resident in LDS");
    if (p.nchunks != N::NCH) return fail(EVD_E_INVALID, "evd_voxel: packed stream has %d chunks, kernel expects %d", p.nchunks, N::NCH);
    if (p.feature || p.act) return fail(EVD_E_INVALID, "evd_voxel: the level's inference pass writes raw only");
    int cus = 256;
    { int dev = 0, v = 0; if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v; }
    const long groups = cdiv(p.nsamp, (long)C::SAMPLES);
    if constexpr (PREC == EVD_PREC_F16X3) {
        // the coarse level of an f16c RENDER (p.rev_trig, evd_voxel_api.hip): sines as in that mode's fine-level kernel; EVD_COARSE_TRIG=exact (developer switch): the polynomial
        static const bool exact = [] { const char* e = getenv("EVD_COARSE_TRIG"); return e && !strcmp(e, "exact"); }();
        if (p.rev_trig && !exact) {
            EVD_SET_MAX_LDS((&k_voxel_mlp_resident<PREC, HD, G, FT, NT, false, false, true>), lds);
            hipLaunchKernelGGL((k_voxel_mlp_resident<PREC, HD, G, FT, NT, false, false, true>), dim3((unsigned)(groups < cus ? groups : cus)), dim3(NT), lds, st, p);
            EVD_LAUNCH_CHECK();
            return EVD_OK;
        }
    }
    EVD_SET_MAX_LDS((&k_voxel_mlp_resident<PREC, HD, G, FT, NT>), lds);
    hipLaunchKernelGGL((k_voxel_mlp_resident<PREC, HD, G, FT, NT>), dim3((unsigned)(groups < cus ? groups : cus)), dim3(NT), lds, st, p);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}
int launch_voxel_coarse_pipe_bf16(const VoxMlpParams& p, hipStream_t st);
int launch_voxel_coarse_pipe_f16(const VoxMlpParams& p, hipStream_t st);
int launch_voxel_coarse_pipe_f16x3(const VoxMlpParams& p, hipStream_t st);

constexpr bool voxel_pipe_built(int prec, int HD, int G, int FT) {
    return (prec == EVD_PREC_BF16 || prec == EVD_PREC_F16 || prec == EVD_PREC_F16X3) && HD == 256 && G == 128 && FT == 64;
}
int launch_voxel_pipe_bf16(bool feat, const VoxMlpParams& p, hipStream_t st);
int launch_voxel_pipe_f16(bool feat, const VoxMlpParams& p, hipStream_t st);
int launch_voxel_pipe_f16x3(bool feat, const VoxMlpParams& p, hipStream_t st);

}  // namespace evd
