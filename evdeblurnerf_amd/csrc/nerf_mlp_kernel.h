// Fused NeRF backbone: pts = o + d z -> positional encodings -> D x W MLP (skip + view branch) -> raw.
// reference: networks/nerf.py:46-72 (mlpforward), :131-162 (eval), networks/embedding.py:88-98, renderer.py:180.
//
// Design (nerf_mlp.h: fragment/permutation contract; mlp_pipe.h: the software pipeline):
//   * one wavefront owns NS x 32 samples for the whole network; activations never leave registers (the MFMA D
//     fragment of a layer IS the B fragment of the next one),
//   * weights are the A operand: one pre-permuted stream of MFMA fragments per precision, LDS-DMA'd into a
//     4-slot ring shared by the workgroup (one barrier per 16 KiB chunk), read back with ds_read_b128,
//   * epilogues, bias loads and fragment prefetches are interleaved between the MFMAs, across layers,
//   * three arithmetic modes share the code: bf16 (32x32x16), split-float16 x3 (32x32x16, hi/lo operands,
//     2^11-scaled cross terms in a second accumulator) and exact float32 (32x32x2).
// The layer table is static (W, D, skip are template parameters): the whole network is one straight-line
// instruction stream.  This header is included by one .hip file per precision (parallel compilation).
#pragma once

#include "mlp_pipe.h"

namespace evd {

// static layer table of the NeRF network: width W (T tiles, KS k-steps), depth D, skip index SKIP (the layer fed by
// cat([input_pts, h]) is SKIP + 1, nerf.py:137-138), configuration C, FEAT = feature rows wanted
template <class C, int W, int D, int SKIP, bool FEAT, bool TRAIN = false> struct NerfNet {
    static constexpr int T = W / 32, KS = W / 16, FPC = C::FPC, PD = C::PD;
    static constexpr int HG = nerf_group(C::PRECISION);            // tile-group size (also the packer's fragment order)
    static constexpr int VG = (T / 2) % HG == 0 ? HG : 1;          // ... of the views layer (T / 2 tiles)
    static constexpr int PDH = KS - 2 * HG;                        // k-step at which a hidden layer's pending group lands
    static constexpr int CH_L0 = cceil(T * PE_KS, FPC), CH_HID = cceil(T * KS, FPC), CH_WIDE = cceil(T * (KS + PE_KS), FPC);
    static constexpr bool is_wide(int l) { return l - 1 == SKIP; }
    // training kernels: slot of hidden layer l's output in the activation store, and of its last (pending) group
    static constexpr int oslot(int l) { return TRAIN ? astore::H0 + 16 * l : -1; }
    static constexpr int pslot(int l) { return TRAIN ? astore::H0 + 16 * l + 2 * (T - HG) : -1; }
    static constexpr int mslot(int l) { return TRAIN ? astore::M_H0 + l : -1; }      // bit mask of hidden layer l's ReLU pattern
    static constexpr int PF0 = 2 * (T - HG);                                        // first pending fragment of a T-tile layer
    static_assert(!TRAIN || (W == 256 && D == 8), "the activation store is laid out for the 8 x 256 network");
    static constexpr int chunk0(int l) {                           // first chunk of hidden layer l (l == D: the heads)
        int c = CH_L0;
        for (int i = 1; i < l; ++i) c += is_wide(i) ? CH_WIDE : CH_HID;
        return c;
    }
    // layer 0: PE(pts) -> W
    typedef LayerDesc<PE_KS, T, HG, true, false, 0, 0, true, 0, 0, 0, false, 0, -1, FEAT && D == 1, HG, oslot(0), -1, mslot(0)> L0;
    static constexpr int par(int l) { return (L0::PAR_OUT + (l - 1) * (T / HG)) & 1; }   // accumulator parity entering hidden layer l
    // hidden layer l (1 .. D-1); the skip layer's k-step order is [h_0..h_{PDH-1} | pe_0..3 | h_PDH..h_{KS-1}]
    template <int l> using Hidden = std::conditional_t<
        is_wide(l),
        LayerDesc<KS + PE_KS, T, HG, true, false, chunk0(l), 0, true, 0, par(l), HG, true, KS + PE_KS - 2 * HG, -1, FEAT && l == D - 1, HG, oslot(l), pslot(l - 1), mslot(l), mslot(l - 1), PF0>,
        LayerDesc<KS, T, HG, true, false, chunk0(l), 0, true, 0, par(l), HG, true, PDH, -1, FEAT && l == D - 1, HG, oslot(l), pslot(l - 1), mslot(l), mslot(l - 1), PF0>>;
    // heads (nerf.py:144-157): alpha_linear, feature_linear, views_linears.0 on cat([feature, PE(dir)]), rgb_linear
    static constexpr int CH_H = chunk0(D);
    typedef LayerDesc<KS, 1, 1, false, true, CH_H, 0, false, 0, par(D), HG, true, PDH, FEAT ? T - HG : -1, false, HG, -1, pslot(D - 1), -1, mslot(D - 1), PF0> Alpha;
    static constexpr int F1 = KS;
    typedef LayerDesc<KS, T, HG, false, false, CH_H, F1, false, F1 % PD, Alpha::PAR_OUT, 0, false, 0, -1, FEAT, VG, TRAIN ? astore::F : -1> Feature;
    static constexpr int F2 = F1 + T * KS;
    typedef LayerDesc<KS + PEV_KS, T / 2, VG, true, false, CH_H, F2, false, F2 % PD, Feature::PAR_OUT, HG, false, PDH, FEAT ? T - HG : -1, false, 1, TRAIN ? astore::HV : -1, TRAIN ? astore::F + 2 * (T - HG) : -1, TRAIN ? astore::M_HV : -1> Views;
    static constexpr int F3 = F2 + (T / 2) * (KS + PEV_KS);
    typedef LayerDesc<KS / 2, 1, 1, false, true, CH_H, F3, true, F3 % PD, Views::PAR_OUT, VG, true, KS / 2 - 2 * VG, -1, false, 0, -1, TRAIN ? astore::HV + 2 * (T / 2 - VG) : -1, -1, TRAIN ? astore::M_HV : -1, 2 * (T / 2 - VG)> Rgb;
    static constexpr int NCH = CH_H + cceil(F3 + KS / 2, FPC);     // chunks of the whole stream
    static_assert(D >= 1 && T % HG == 0 && (T * PE_KS) % PD == 0 && (T * KS) % PD == 0, "fragment counts must keep the prefetch ring phase");
};

// everything a layer call needs, bundled so that the hidden-layer recursion stays readable
template <class C, class N, bool FEAT, bool TRAIN = false> struct NerfCtx {
    typedef typename C::O::B B;
    PStream<C, FEAT || TRAIN, N::NCH> st;
    Pipe<C> pp;
    B buf[2][C::NS][N::KS];         // activations ping-pong: layer l writes buf[l & 1]
    B* stash;                       // this lane's slot of the wavefront's positional-encoding stash
    const float* bias;              // LDS bias block
    int lane;
    float* const* frow_before;      // feature rows when "before_linear" is wanted, else nulls
    float* const* frow_after;       // ... "after_linear"
    float* const* nofrow;
    char* const* act;               // training: this lane's 16 bytes in fragment 0 of each sample tile's activation store
};

template <class C, class N, bool FEAT, bool TRAIN, int l, int D> struct HiddenLoop {
    static __device__ __forceinline__ void run(NerfCtx<C, N, FEAT, TRAIN>& cx) {
        typedef typename C::O::B B;
        typedef typename N::template Hidden<l> L;
        constexpr int NS = C::NS, KS = N::KS, T = N::T;
        const float* lb = cx.bias + l * T * 32;
        float* const* fr = (FEAT && l == D - 1) ? cx.frow_before : cx.nofrow;
        if constexpr (N::is_wide(l)) {
            B wide[NS][KS + PE_KS];
            const B* sp = cx.stash;
            asm volatile("" : "+v"(sp));    // opaque: else hipcc reuses layer 0's loads and keeps 16 registers live for 5 layers
#pragma unroll
            for (int s = 0; s < NS; ++s) {
#pragma unroll
                for (int j = 0; j < N::PDH; ++j) wide[s][j] = cx.buf[(l - 1) & 1][s][j];
#pragma unroll
                for (int j = 0; j < PE_KS; ++j) wide[s][N::PDH + j] = sp[(s * C::STASH_FRAGS + j) * 64];
            }
            pipe_layer<C, L, decltype(cx.st), KS, TRAIN>(cx.st, cx.pp, wide, cx.buf[l & 1], nullptr, lb, cx.lane, fr, cx.act);
        } else {
            pipe_layer<C, L, decltype(cx.st), KS, TRAIN>(cx.st, cx.pp, cx.buf[(l - 1) & 1], cx.buf[l & 1], nullptr, lb, cx.lane, fr, cx.act);
        }
        if constexpr (l + 1 < D) HiddenLoop<C, N, FEAT, TRAIN, l + 1, D>::run(cx);
    }
};

template <int PREC, int W, int D, int SKIP, int NS, int NT, bool FEAT, bool TRAIN = false, bool HI_ONLY = false>
__global__ __launch_bounds__(NT, NT / 256) void k_nerf_mlp(const MlpParams p) {
    typedef PipeCfg<PREC, NS, NT, PIPE_CB, HI_ONLY> C;
    typedef typename C::O O;
    typedef typename O::B B;
    typedef NerfNet<C, W, D, SKIP, FEAT, TRAIN> N;
    constexpr int T = N::T, KS = N::KS;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    pipe_fp16_saturate<PREC>();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, h = lane >> 5;
    NerfCtx<C, N, FEAT, TRAIN> cx;
    cx.st.start_issue(p.wstream, smem, tid);
    float* bias = reinterpret_cast<float*>(smem + C::RING);
    {   // bias block -> LDS, all loads of a thread in flight together
        constexpr int NB = (C::BIAS_FLOATS / 4 + NT - 1) / NT;
        f32x4 bv[NB];
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int i = (tid + q * NT) * 4;
            bv[q] = i < p.nbias ? *reinterpret_cast<const f32x4*>(p.bias + i) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int i = (tid + q * NT) * 4;
            if (i < p.nbias) *reinterpret_cast<f32x4*>(bias + i) = bv[q];
        }
    }
    cx.stash = reinterpret_cast<B*>(smem + C::RING + C::BIAS_FLOATS * 4 + wave * C::STASH_PER_WAVE) + lane;
    cx.bias = bias;
    cx.lane = lane;

    // both positional encodings straight into B-fragment order, parked in this wavefront's LDS stash until layer 0,
    // the skip layer and the views layer read them back (nothing of the prologue stays in registers)
    long sidx[NS];
    bool valid[NS];
    char* actl[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const long smp = (long)blockIdx.x * C::SAMPLES + wave * (NS * 32) + s * 32 + n;
        actl[s] = TRAIN ? p.act + (smp >> 5) * astore::tile_bytes(C::STORE_PREC) + lane * 16 : nullptr;
        valid[s] = smp < p.nsamp;
        sidx[s] = valid[s] ? smp : p.nsamp - 1;
        const long ray = sidx[s] / p.S;
        const float* rb = p.ray_batch + ray * p.ncol;
        const float zv = p.z[sidx[s]];
        float pts[3], vd[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            pts[c] = __fadd_rn(rb[c], __fmul_rn(rb[3 + c], zv));   // renderer.py:180
            vd[c] = rb[8 + c];
        }
        B pe[PE_KS], pev[PEV_KS];
        encode_pairs<C, PE_L, PE_KS>(pts, h, pe);
        encode_pairs<C, PE_LV, PEV_KS>(vd, h, pev);
#pragma unroll
        for (int j = 0; j < PE_KS; ++j) cx.stash[(s * C::STASH_FRAGS + j) * 64] = pe[j];
#pragma unroll
        for (int j = 0; j < PEV_KS; ++j) cx.stash[(s * C::STASH_FRAGS + PE_KS + j) * 64] = pev[j];
        if constexpr (TRAIN) {
#pragma unroll
            for (int j = 0; j < PE_KS; ++j) pipe_act_store<C>(actl[s], astore::PE + j, pe[j]);
#pragma unroll
            for (int j = 0; j < PEV_KS; ++j) pipe_act_store<C>(actl[s], astore::DIR + j, pev[j]);
        }
    }
    cx.act = actl;
    float* frow[NS];
    float* nofrow[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        frow[s] = (FEAT && p.feature && valid[s]) ? p.feature + sidx[s] * W : nullptr;
        nofrow[s] = nullptr;
    }
    cx.frow_before = p.feature_kind == 2 ? frow : nofrow;     // "before_linear": output of the last hidden layer
    cx.frow_after = p.feature_kind == 1 ? frow : nofrow;      // "after_linear": feature_linear output
    cx.nofrow = nofrow;

    cx.st.start_wait();
#ifdef EVD_PIPE_SHIFT
#pragma unroll
    for (int i = 0; i < EVD_PIPE_SHIFT; ++i) asm volatile("s_nop 0");
#endif
    pipe_prime<C, typename N::L0>(cx.st, cx.pp, bias, lane);
    {
        B in_pe[NS][PE_KS];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int j = 0; j < PE_KS; ++j) in_pe[s][j] = cx.stash[(s * C::STASH_FRAGS + j) * 64];
        pipe_layer<C, typename N::L0, decltype(cx.st), KS, TRAIN>(cx.st, cx.pp, in_pe, cx.buf[0], nullptr, bias, lane, D == 1 ? cx.frow_before : nofrow, actl);
    }
    if constexpr (D > 1) HiddenLoop<C, N, FEAT, TRAIN, 1, D>::run(cx);

    // heads
    B (&hact)[NS][KS] = cx.buf[(D - 1) & 1];
    const float* lb = bias + D * T * 32;
    float araw[NS][4], rraw[NS][4];
    B none[NS][1];
    pipe_layer<C, typename N::Alpha, decltype(cx.st), 1, TRAIN>(cx.st, cx.pp, hact, none, araw, lb, lane, cx.frow_before, actl);
    lb += 32;
    B vin[NS][KS + PEV_KS];
    pipe_layer<C, typename N::Feature, decltype(cx.st), KS + PEV_KS, TRAIN>(cx.st, cx.pp, hact, vin, nullptr, lb, lane, cx.frow_after, actl);
    lb += T * 32;
    {
        const B* sp = cx.stash;
        asm volatile("" : "+v"(sp));        // opaque: these loads must not be hoisted above the feature layer
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int j = 0; j < PEV_KS; ++j) vin[s][KS + j] = sp[(s * C::STASH_FRAGS + PE_KS + j) * 64];
    }
    B hbuf[NS][KS / 2];
    pipe_layer<C, typename N::Views, decltype(cx.st), KS / 2, TRAIN>(cx.st, cx.pp, vin, hbuf, nullptr, lb, lane, cx.frow_after, actl);
    lb += (T / 2) * 32;
    pipe_layer<C, typename N::Rgb, decltype(cx.st), 1, TRAIN>(cx.st, cx.pp, hbuf, none, rraw, lb, lane, nofrow, actl);

#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (h == 0 && valid[s]) {
            const f32x4 o = {rraw[s][0], rraw[s][1], rraw[s][2], araw[s][0]};   // cat([rgb, alpha]) nerf.py:157
            *reinterpret_cast<f32x4*>(p.raw + sidx[s] * 4) = o;
        }
    }
}

// chunks of the packed stream the kernel <PREC, W, D, SKIP> expects (checked against the packer's count)
template <int PREC, int W, int D, int SKIP> constexpr int nerf_pipe_chunks() {
    return NerfNet<PipeCfg<PREC, 1, 256>, W, D, SKIP, false>::NCH;
}

template <int PREC, int W, int D, int SKIP, int NS, int NT, bool FEAT, bool TRAIN = false, bool HI_ONLY = false>
static int launch_pipe_mlp(const MlpParams& p, hipStream_t st) {
    typedef PipeCfg<PREC, NS, NT, PIPE_CB, HI_ONLY> C;
    // training: the store is tiled in groups of 8 tiles (256 samples, evd_train_api.hip train_tiles) and the backward kernels walk ALL of
    // them, so the forward fills the padding tiles too (clamped copies of the last sample), whatever its own workgroup size
    const long blocks = TRAIN ? cdiv(p.nsamp, 256L) * (256 / C::SAMPLES) : cdiv(p.nsamp, C::SAMPLES);
    static_assert(256 % C::SAMPLES == 0 || !TRAIN, "training workgroups tile the 256-sample groups of the store");
    const size_t lds = C::TOTAL;
    EVD_SET_MAX_LDS((&k_nerf_mlp<PREC, W, D, SKIP, NS, NT, FEAT, TRAIN, HI_ONLY>), lds);
    if (p.nbias > C::BIAS_FLOATS) return fail(EVD_E_INVALID, "evd_nerf_mlp: %d bias floats exceed the LDS bias block", p.nbias);
    if (p.nchunks != NerfNet<C, W, D, SKIP, FEAT>::NCH)
        return fail(EVD_E_INVALID, "evd_nerf_mlp: packed stream has %d chunks, kernel expects %d", p.nchunks, NerfNet<C, W, D, SKIP, FEAT>::NCH);
    if (TRAIN && !p.act) return fail(EVD_E_INVALID, "evd_nerf_mlp: training launch without an activation store");
    hipLaunchKernelGGL((k_nerf_mlp<PREC, W, D, SKIP, NS, NT, FEAT, TRAIN, HI_ONLY>), dim3((unsigned)blocks), dim3(NT), lds, st, p);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

// pipelined kernels are built for these static networks; everything else runs on the generic kernel (kernel_nerf_mlp.hip)
constexpr bool nerf_pipe_built(int prec, int W, int D, int skip) {
    return (prec == EVD_PREC_BF16 || prec == EVD_PREC_F16 || prec == EVD_PREC_F16X3) && W == 256 && D == 8 && skip == 4;
}
// p.act set: the training variant (saves the activations the backward kernels need)
int launch_nerf_pipe_bf16(bool feat, const MlpParams& p, hipStream_t st);
int launch_nerf_pipe_f16(bool feat, const MlpParams& p, hipStream_t st);
int launch_nerf_pipe_f16x3(bool feat, const MlpParams& p, hipStream_t st);

}  // namespace evd
