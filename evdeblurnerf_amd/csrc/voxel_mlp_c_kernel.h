// PDRF fine-level network (hidden 256, geo 128, 64 feature channels in; reference networks/pdrf/voxnerf.py:210-221,240-254 with the
// blurfactory dimensions) in the COMPENSATED float16 mode (EVD_PREC_F16C): sigma net 127 -> 256 -> 1 + 128, colour net 155 -> 256 -> 256 -> 3
// (sigmoid).  Machinery and arithmetic: mlp_pipe_c.h (one float16 MFMA product + two block-scaled fp6 products of the operands' rounding
// residuals, one wavefront of 32 samples per SIMD).  This is the mode the shipped (mode='c2f') configurations are rendered in when the
// 1e-4 RGB bound has to hold on trained weights: the single-product float16 mode measures 3.8e-4 there (tools/trained_c2f.py).
// No feature rows; the coarse 64-wide level runs in EVD_PREC_F16X3 next to it (evd_voxel_api.hip).
// TRAIN variant (round 4): the same arithmetic, and every completed input / hidden / geo block also goes to the activation store of the
// single-product float16 mode (voxel_mlp_kernel.h VStore: float16 fragments in that mode's arrangement + the ReLU patterns as bit masks,
// both taken from the COMPENSATED pre-activations), so that the level's backward is that mode's dgrad / wgrad chain unchanged.  This is
// the training forward that holds the reference's float32 numbers (run_nerf.py:593-601 trains in float32): rendered colours, the
// resampled positions and the ReLU patterns are the float32 ones to ~2^-15 instead of 2^-11.
#pragma once

#include "mlp_pipe_c.h"
#include "voxel.h"
#include "voxel_mlp_kernel.h"

namespace evd {

// static layer table of one level: HD hidden width, G geo channels (a multiple of 64), FT feature channels in (64: one input block)
template <int HD, int G, int FT, bool TRAIN = false> struct VoxNetC {
    typedef VStore<HD, G, FT> VS;
    static constexpr int slot(int s) { return TRAIN ? s : -1; }
    static constexpr int T = HD / 32, KB = HD / 64, GT = G / 32, GB = G / 64;
    static_assert(FT == 64 && PE_KS == 4 && PEV_KS == 2 && T % 2 == 0 && G % 64 == 0 && KB >= 2, "block structure of the inputs");
    // sigma_net.0 on cat([fts, PE(pts)]) (voxnerf.py:214): blocks [fts | pe]
    typedef CLayer<2, 4, T, 2, true, false, 0, 0, 0, false, 0, 1, slot(VS::HID), -1, slot(VS::M_HID)> L0;
    // sigma_net.1 row 0 = sigma (float32 out); drains L0's last group into the hidden block KB - 1
    typedef CLayer<KB, 4, 1, 1, false, true, L0::NCHUNKS, L0::PAR_OUT, 2, true, KB - 1, 2, -1, slot(VS::HID + 4 * (KB - 1)), -1, slot(VS::M_HID)> Sigma;
    // sigma_net.1 rows 1..G = geo features, no activation (voxnerf.py:221)
    typedef CLayer<KB, 4, GT, 2, false, false, Sigma::CHUNK0 + Sigma::NCHUNKS, Sigma::PAR_OUT, 0, false, 0, 2, slot(VS::GEO)> Geo;
    // color_net.0 on cat([geo, PE(dirs)]) (voxnerf.py:248): blocks [geo_0 .. geo_{GB-1} | dir (2 k-steps)]; drains geo's last group into block GB - 1
    typedef CLayer<GB + 1, 2, T, 2, true, false, Geo::CHUNK0 + Geo::NCHUNKS, Geo::PAR_OUT, 2, false, GB - 1, 2, slot(VS::C0), slot(VS::GEO + 4 * (GB - 1)),
                   slot(VS::M_C0)> C0;
    typedef CLayer<KB, 4, T, 2, true, false, C0::CHUNK0 + C0::NCHUNKS, C0::PAR_OUT, 2, true, KB - 1, 1, slot(VS::C1), slot(VS::C0 + 4 * (KB - 1)), slot(VS::M_C1),
                   slot(VS::M_C0)> C1;
    typedef CLayer<KB, 4, 1, 1, false, true, C1::CHUNK0 + C1::NCHUNKS, C1::PAR_OUT, 2, true, KB - 1, 0, -1, slot(VS::C1 + 4 * (KB - 1)), -1, slot(VS::M_C1)> C2;
    static constexpr int NCH = C2::CHUNK0 + C2::NCHUNKS;
    // LDS bias / row-scale image in stream order (the sigma net has no biases: zeros)
    static constexpr int B_SIG = T * 32, B_GEO = B_SIG + 32, B_C0 = B_GEO + GT * 32, B_C1 = B_C0 + T * 32, B_C2 = B_C1 + T * 32, B_END = B_C2 + 32;
    static constexpr int NTILES = B_END / 32;
    static_assert(B_END <= CCfg::BIAS_WORDS / 2, "bias / row-scale block");
};

// one input block from 64 float32 values of a row: B position q = 8 j + e of lane half h <-> value 16 j + 8 h + e (the natural k-step order
// of evd_voxel_api.hip in0_col), all three representations
__device__ __forceinline__ void c_block_from_row(const float* f, XBlk& out) {
    f32x16 r[2];               // element i = 8 j + e of the block is pair i / 2, member i & 1 (mlp_pipe_c.h c_drain_pair)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(f + 16 * j), b = *reinterpret_cast<const f32x4*>(f + 16 * j + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int i = 8 * j + e;
            r[i & 1][i >> 1] = e < 4 ? a[e] : b[e - 4];
        }
    }
    unsigned m = 0u;
#pragma unroll
    for (int v = 0; v < 16; ++v) c_drain_pair<false>(r[0], r[1], v, out, m);
    c_finish(out, m, r[0], r[1]);
}

// ... from the row's 32 values of this lane half already in registers (a[2 j], a[2 j + 1] = values 16 j + 8 h + 0..7)
__device__ __forceinline__ void c_block_from_regs(const f32x4 (&a)[8], XBlk& out) {
    f32x16 r[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int i = 8 * j + e;
            r[i & 1][i >> 1] = e < 4 ? a[2 * j][e] : a[2 * j + 1][e - 4];
        }
    }
    unsigned m = 0u;
#pragma unroll
    for (int v = 0; v < 16; ++v) c_drain_pair<false>(r[0], r[1], v, out, m);
    c_finish(out, m, r[0], r[1]);
}

template <int HD, int G, int FT, bool TRAIN>
__global__ __launch_bounds__(CCfg::NT, 1) void k_voxel_mlp_c(const VoxMlpParams p) {
    typedef VoxNetC<HD, G, FT, TRAIN> N;
    typedef typename N::VS VS;
    typedef CStream<N::NCH> ST;
    constexpr int KB = N::KB, GB = N::GB;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    pipe_fp16_saturate<EVD_PREC_F16>();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, h = lane >> 5;
    ST st;
    st.start_issue(p.wstream, smem, tid);
    float* bias = reinterpret_cast<float*>(smem + CCfg::RING);
    // bias image (zeros for the sigma net, then the colour-net biases: p.bias = 512 zeros + colour biases, evd_voxel_api.hip) and row scales
    for (int i = tid; i < N::B_END; i += CCfg::NT) {
        bias[i] = i < N::B_C0 ? 0.f : p.bias[512 + (i - N::B_C0)];
        reinterpret_cast<unsigned*>(bias)[CCfg::BIAS_WORDS / 2 + i] = p.wscale[i];
    }
    unsigned bias_off = lds_offset_of(bias);
    asm volatile("" : "+v"(bias_off));     // opaque base: the bias / row-scale reads then take immediate offsets
    const lds_f32_p lbias = (lds_f32_p)(unsigned long)bias_off;

    // persistent workgroups, one per CU (nerf_mlp_c_kernel.h: the hand-over between two workgroups of a one-workgroup-per-CU kernel is
    // uncovered; a pass of this network is only ~19 us long)
    // (TRAIN: the store is tiled in groups of 8 sample tiles and the backward walks all of them -- the padding tiles are written too)
    const long ntile = TRAIN ? (p.nsamp + 255) / 256 * (256 / CCfg::SAMPLES) : (p.nsamp + CCfg::SAMPLES - 1) / CCfg::SAMPLES;
#ifdef EVD_VC_ABL       // developer ablation (tools/dev/voxel_c_prologue_ablation.sh; results wrong by construction): bit 0 -- the two positional encodings,
    XBlk in0[2], pev;   // bit 1 -- the feature block too, are built on the workgroup's FIRST pass only and kept: what hiding that work could gain at most
#endif
    for (long tile = blockIdx.x;;) {
#ifdef EVD_C_STAMP      // developer build (tools/dev/stamp_voxel_c.py): shader-clock stamps of this pass, written by lanes 0..2 in place of their samples
    long long ts[10];
    ts[0] = __builtin_readcyclecounter();
    st.tw = 0; st.tb = 0;
#define EVD_VSTAMP(i) ts[i] = __builtin_readcyclecounter()
#else
#define EVD_VSTAMP(i)
#endif
    const long smp = tile * CCfg::SAMPLES + wave * 32 + n;
    const bool valid = smp < p.nsamp;
    const long sidx = valid ? smp : p.nsamp - 1;
#ifndef EVD_VC_ABL
    XBlk in0[2], pev;
#endif
    {
        // All loads first, the 256-byte feature row (Infinity Cache / HBM: the gather kernel wrote 134 MB of them) LAST in the queue and
        // FIRST in need of time: the two positional encodings (~500 VALU instructions on six floats) run while it is in flight, the
        // feature block is built behind them.  (Stamps, profiles/r04_voxel_c_stamps.log: inputs + encode + the wait in front of the first
        // chunk were 10.3 k of a pass's 42 k cycles with the feature block built first.)
        float pts[3], vd[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            pts[c] = p.pts[sidx * 3 + c];
            vd[c] = p.viewdirs[(sidx / p.S) * p.vd_stride + c];
        }
        const float* f = p.fts + sidx * (long)p.ft_stride + 8 * h;
        f32x4 fr[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            fr[2 * j] = *reinterpret_cast<const f32x4*>(f + 16 * j);
            fr[2 * j + 1] = *reinterpret_cast<const f32x4*>(f + 16 * j + 4);
        }
        asm volatile("" ::: "memory");       // the loads stay in front of the encodings
#ifdef EVD_VC_ABL
        if (tile == blockIdx.x) {
            c_encode<PE_L, PE_KS>(pts, h, in0[1]);
            c_encode<PE_LV, PEV_KS>(vd, h, pev);
        }
        if (tile == blockIdx.x || !(EVD_VC_ABL & 2)) c_block_from_regs(fr, in0[0]);
#else
        c_encode<PE_L, PE_KS>(pts, h, in0[1]);
        c_encode<PE_LV, PEV_KS>(vd, h, pev);
        c_block_from_regs(fr, in0[0]);
#endif
    }
    CAct act{};
    if constexpr (TRAIN) {
        const long t32 = __builtin_amdgcn_readfirstlane((int)(tile * (CCfg::SAMPLES / 32) + wave));      // this wavefront's 32-sample tile (< 2^31)
        act.base = p.act + t32 * VS::TILE_BYTES;
        act.voff = lane * 16;
        // the network's inputs, in the float16 mode's own fragment order (voxel_mlp_kernel.h)
        c_store_input<4>(act, VS::IN0, in0[0]);
        c_store_input<PE_KS>(act, VS::IN0 + 4, in0[1]);
        c_store_input<PEV_KS>(act, VS::DIRPE, pev);
    }
    EVD_VSTAMP(1);
    st.start_wait();
    CPipe pp;
    c_prime<typename N::L0>(st, pp, lbias, lane);
    EVD_VSTAMP(2);
    XBlk hid[KB], none[1];
    c_layer<typename N::L0, typename N::Sigma, ST, 2, KB, TRAIN>(st, pp, in0, hid, nullptr, lbias, lane, act);
    EVD_VSTAMP(3);
    float sig[4], col[4];
    c_layer<typename N::Sigma, typename N::Geo, ST, KB, 1, TRAIN>(st, pp, hid, none, sig, lbias + N::B_SIG, lane, act);
    EVD_VSTAMP(4);
    XBlk cin[GB + 1];
    c_layer<typename N::Geo, typename N::C0, ST, KB, GB + 1, TRAIN>(st, pp, hid, cin, nullptr, lbias + N::B_GEO, lane, act);
    EVD_VSTAMP(5);
    cin[GB] = pev;
    XBlk c0[KB], c1[KB];
    c_layer<typename N::C0, typename N::C1, ST, GB + 1, KB, TRAIN>(st, pp, cin, c0, nullptr, lbias + N::B_C0, lane, act);
    EVD_VSTAMP(6);
    c_layer<typename N::C1, typename N::C2, ST, KB, KB, TRAIN>(st, pp, c0, c1, nullptr, lbias + N::B_C1, lane, act);
    EVD_VSTAMP(7);
    c_layer<typename N::C2, void, ST, KB, 1, TRAIN>(st, pp, c1, none, col, lbias + N::B_C2, lane, act);
    EVD_VSTAMP(8);
    const bool more = tile + gridDim.x < ntile;
    if (more) st.restart_issue();               // behind the barrier of the stream's last chunk: all slots are free

    if (h == 0 && valid) {
        f32x4 o;
        o[0] = sig[0];
#pragma unroll
        for (int c = 0; c < 3; ++c) o[1 + c] = 1.f / (1.f + expf(-col[c]));      // torch.sigmoid(h) voxnerf.py:252
#ifdef EVD_C_STAMP
        ts[9] = __builtin_readcyclecounter();
        if (lane == 0) o = f32x4{(float)(ts[1] - ts[0]), (float)(ts[2] - ts[1]), (float)(ts[3] - ts[2]), (float)(ts[4] - ts[3])};
        if (lane == 1) o = f32x4{(float)(ts[5] - ts[4]), (float)(ts[6] - ts[5]), (float)(ts[7] - ts[6]), (float)(ts[8] - ts[7])};
        if (lane == 2) o = f32x4{(float)(ts[9] - ts[8]), (float)st.tw, (float)st.tb, -7.f};
        if (lane == 3) o = f32x4{(float)(ts[0] & 0xffffff), (float)((ts[0] >> 24) & 0xffffff), (float)blockIdx.x, (float)tile};
#endif
        *reinterpret_cast<f32x4*>(p.raw + sidx * 4) = o;
    }
    if (!more) break;
    tile += gridDim.x;
    }
}

template <int HD, int G, int FT, bool TRAIN = false>
static int launch_voxel_c(const VoxMlpParams& p, hipStream_t st) {
    typedef VoxNetC<HD, G, FT, TRAIN> N;
    const long tiles = TRAIN ? cdiv(p.nsamp, 256L) * (256 / CCfg::SAMPLES) : cdiv(p.nsamp, CCfg::SAMPLES);
    const long blocks = cmin_l(tiles, (long)c_persistent_blocks());
    const size_t lds = CCfg::TOTAL;
    EVD_SET_MAX_LDS((&k_voxel_mlp_c<HD, G, FT, TRAIN>), lds);
    if (p.nchunks != N::NCH) return fail(EVD_E_INVALID, "evd_voxel (f16c): packed stream has %d chunks, kernel expects %d", p.nchunks, N::NCH);
    if (!p.wscale) return fail(EVD_E_INVALID, "evd_voxel (f16c): no row scales");
    if (p.feature) return fail(EVD_E_INVALID, "evd_voxel (f16c): feature rows are not built in this mode (use EVD_PREC_F16X3)");
    if (TRAIN != (p.act != nullptr)) return fail(EVD_E_INVALID, "evd_voxel (f16c): the activation store goes with the training launch");
    hipLaunchKernelGGL((k_voxel_mlp_c<HD, G, FT, TRAIN>), dim3((unsigned)blocks), dim3(CCfg::NT), lds, st, p);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

constexpr bool voxel_c_built(int HD, int G, int FT) { return HD == 256 && G == 128 && FT == 64; }

}  // namespace evd
