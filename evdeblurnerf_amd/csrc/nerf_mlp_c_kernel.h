// Fused NeRF backbone in the compensated float16 mode (EVD_PREC_F16C): pts = o + d z -> positional encodings -> 8 x 256 MLP (skip +
// view branch) -> raw.  reference: networks/nerf.py:46-72 (mlpforward), :131-162 (eval), networks/embedding.py:88-98, renderer.py:180.
// Machinery and arithmetic: mlp_pipe_c.h.  Built for the reference network (netdepth 8, netwidth 256, skips [4]); no feature rows.
// TRAIN variant (round 4): + the activation store of the single-product float16 mode (nerf_mlp.h astore: float16 fragments in that
// mode's arrangement, ReLU patterns as bit masks from the COMPENSATED pre-activations); the backward is that mode's.
#pragma once

#include "mlp_pipe_c.h"
#include "wave_ops.h"

namespace evd {

// static layer table: width W (T tiles, KB k-blocks), depth D, skip index SKIP
template <int W, int D, int SKIP, bool TRAIN = false> struct NerfNetC {
    static constexpr int T = W / 32, KB = W / 64;
    static constexpr int slot(int s) { return TRAIN ? s : -1; }
    static constexpr int oslot(int l) { return slot(astore::H0 + 16 * l); }                       // hidden layer l's output ...
    static constexpr int pslot(int l) { return slot(astore::H0 + 16 * l + 4 * (KB - 1)); }        // ... its last (pending) block
    static constexpr int mslot(int l) { return slot(astore::M_H0 + l); }
    static_assert(!TRAIN || (W == 256 && D == 8), "the activation store is laid out for the 8 x 256 network");
    static_assert(PE_KS == 4 && PEV_KS == 2 && T % 2 == 0 && KB >= 2, "block structure of the encodings and the hidden width");
    static constexpr bool is_wide(int l) { return l - 1 == SKIP; }
    typedef CLayer<1, 4, T, 2, true, false, 0, 0, 0, false, 0, 2, oslot(0), -1, mslot(0)> L0;
    static constexpr int CH_HID = CLayer<KB, 4, T, 2, true, false, 0, 0, 2, true, KB - 1, 2>::NCHUNKS;
    static constexpr int CH_WIDE = CLayer<KB + 1, 4, T, 2, true, false, 0, 0, 2, true, KB, 2>::NCHUNKS;
    static constexpr int chunk0(int l) {                           // first chunk of hidden layer l (l == D: the heads)
        int c = L0::NCHUNKS;
        for (int i = 1; i < l; ++i) c += is_wide(i) ? CH_WIDE : CH_HID;
        return c;
    }
    static constexpr int PARH = L0::PAR_OUT;                       // T / 2 groups per hidden layer: the parity is kept when T / 2 is even
    static_assert((T / 2) % 2 == 0, "accumulator parity");
    // hidden layer l (1 .. D-1); the skip layer's blocks are [h_0 .. h_{KB-2} | pe | h_{KB-1}]
    template <int l> using Hidden = std::conditional_t<is_wide(l),
        CLayer<KB + 1, 4, T, 2, true, false, chunk0(l), PARH, 2, true, KB, l == D - 1 ? 1 : 2, oslot(l), pslot(l - 1), mslot(l), mslot(l - 1), 4 * (KB - 1)>,
        CLayer<KB, 4, T, 2, true, false, chunk0(l), PARH, 2, true, KB - 1, l == D - 1 ? 1 : 2, oslot(l), pslot(l - 1), mslot(l), mslot(l - 1)>>;
    // heads (nerf.py:144-157): alpha_linear, feature_linear, views_linears.0 on cat([feature, PE(dir)]), rgb_linear
    typedef CLayer<KB, 4, 1, 1, false, true, chunk0(D), PARH, 2, true, KB - 1, 2, -1, pslot(D - 1), -1, mslot(D - 1)> Alpha;
    typedef CLayer<KB, 4, T, 2, false, false, Alpha::CHUNK0 + Alpha::NCHUNKS, Alpha::PAR_OUT, 0, false, 0, 2, slot(astore::F)> Feature;
    typedef CLayer<KB + 1, 2, T / 2, 2, true, false, Feature::CHUNK0 + Feature::NCHUNKS, Feature::PAR_OUT, 2, false, KB - 1, 1, slot(astore::HV),
                   slot(astore::F + 4 * (KB - 1)), slot(astore::M_HV)> Views;
    typedef CLayer<KB / 2, 4, 1, 1, false, true, Views::CHUNK0 + Views::NCHUNKS, Views::PAR_OUT, 2, true, KB / 2 - 1, 0, -1, slot(astore::HV + 4 * (KB / 2 - 1)), -1,
                   slot(astore::M_HV)> Rgb;
    static constexpr int NCH = Rgb::CHUNK0 + Rgb::NCHUNKS;
    static constexpr int NTILES = D * T + 1 + T + T / 2 + 1;
    static_assert(D >= 2 && NTILES * 32 <= CCfg::BIAS_WORDS / 2, "bias / row-scale block");
};

template <class N, class ST> struct NerfCtxC {
    ST st;
    CPipe pp;
    XBlk buf[2][N::KB];             // activations ping-pong: layer l writes buf[l & 1]
    XBlk pe, pev;                   // the two positional encodings as input blocks (registers: one wavefront per SIMD has room)
    lds_f32_p bias;                 // LDS bias block
    int lane;
    CAct act;                       // training kernels: this wavefront's tile of the activation store
};

template <class N, class ST, int l, int D, bool TRAIN = false> struct HiddenLoopC {
    static __device__ __forceinline__ void run(NerfCtxC<N, ST>& cx) {
        typedef typename N::template Hidden<l> L;
        constexpr int KB = N::KB, T = N::T;
        lds_f32_p lb = cx.bias + l * T * 32;
        if constexpr (l + 1 < D) {
            typedef typename N::template Hidden<l + 1> NX;
            step<L, NX>(cx, lb);
            HiddenLoopC<N, ST, l + 1, D, TRAIN>::run(cx);
        } else {
            step<L, typename N::Alpha>(cx, lb);
        }
    }
    template <class L, class NX> static __device__ __forceinline__ void step(NerfCtxC<N, ST>& cx, lds_f32_p lb) {
        constexpr int KB = N::KB;
        if constexpr (N::is_wide(l)) {
            XBlk wide[KB + 1];
#pragma unroll
            for (int j = 0; j < KB - 1; ++j) wide[j] = cx.buf[(l - 1) & 1][j];
            wide[KB - 1] = cx.pe;
            c_layer<L, NX, ST, KB + 1, KB, TRAIN>(cx.st, cx.pp, wide, cx.buf[l & 1], nullptr, lb, cx.lane, cx.act);
        } else {
            c_layer<L, NX, ST, KB, KB, TRAIN>(cx.st, cx.pp, cx.buf[(l - 1) & 1], cx.buf[l & 1], nullptr, lb, cx.lane, cx.act);
        }
    }
};

// z of sample i of a ray (renderer.py:163-178; the arithmetic of k_sample_z)
__device__ __forceinline__ float c_z_at(float near, float far, int S, int lindisp, int k) {
    const float t = linspace_at(0.f, 1.f, S, k);
    return lindisp ? 1.f / __fadd_rn(__fmul_rn(1.f / near, __fsub_rn(1.f, t)), __fmul_rn(1.f / far, t))
                   : __fadd_rn(__fmul_rn(near, __fsub_rn(1.f, t)), __fmul_rn(far, t));
}
__device__ __forceinline__ float c_z_sample(float near, float far, int S, int lindisp, int perturb, const float* t_rand, long ray, int i) {
    float zi = c_z_at(near, far, S, lindisp, i);
    if (perturb) {
        const float upper = (i < S - 1) ? __fmul_rn(.5f, __fadd_rn(c_z_at(near, far, S, lindisp, i + 1), zi)) : zi;
        const float lower = (i > 0) ? __fmul_rn(.5f, __fadd_rn(zi, c_z_at(near, far, S, lindisp, i - 1))) : zi;
        zi = __fadd_rn(lower, __fmul_rn(__fsub_rn(upper, lower), t_rand[ray * S + i]));
    }
    return zi;
}

// FUSE: the whole render step of mode='nerf' without importance samples in this one launch (S = 32, 64 or 128: a workgroup's 128
// samples are whole rays): z stratification in the prologue, raw2outputs in the epilogue -- the per-sample (rgb, sigma) never leave
// registers unless `raw` is wanted; transmittance = DPP product scan over a wavefront's 32 samples, carried across the ray's wavefronts
// through 4 words of LDS.
template <int W, int D, int SKIP, bool FUSE, bool TRAIN = false>
__global__ __launch_bounds__(CCfg::NT, 1) void k_nerf_mlp_c(const MlpParams p) {
    typedef NerfNetC<W, D, SKIP, TRAIN> N;
    static_assert(!(FUSE && TRAIN), "the training forward writes raw; the compositing scan runs under autograd");
    typedef CStream<N::NCH> ST;
    constexpr int T = N::T, KB = N::KB;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    pipe_fp16_saturate<EVD_PREC_F16>();
#ifdef EVD_C_STAMP
    const long long t_start = __builtin_readcyclecounter();
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, h = lane >> 5;
    ST st0;
    st0.start_issue(p.wstream, smem, tid);
    float* bias = reinterpret_cast<float*>(smem + CCfg::RING);
    {   // biases and row-scale words -> LDS
        constexpr int NB = (CCfg::BIAS_WORDS / 2 / 4 + CCfg::NT - 1) / CCfg::NT;
        f32x4 bv[NB];
        u32x4 sv[NB];
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int i = (tid + q * CCfg::NT) * 4;
            bv[q] = i < p.nbias ? *reinterpret_cast<const f32x4*>(p.bias + i) : f32x4{0.f, 0.f, 0.f, 0.f};
            sv[q] = i < p.nbias ? *reinterpret_cast<const u32x4*>(p.wscale + i) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int i = (tid + q * CCfg::NT) * 4;
            if (i < p.nbias) {
                *reinterpret_cast<f32x4*>(bias + i) = bv[q];
                *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned*>(bias) + CCfg::BIAS_WORDS / 2 + i) = sv[q];
            }
        }
    }
    unsigned bias_off = lds_offset_of(bias);
    asm volatile("" : "+v"(bias_off));     // opaque base: the bias / row-scale reads then take immediate offsets (the block sits above 64 KiB)
    const lds_f32_p lbias = (lds_f32_p)(unsigned long)bias_off;

    // PERSISTENT workgroups (one per CU, launch_nerf_c): a workgroup walks the 128-sample tiles blockIdx.x, blockIdx.x + gridDim.x, ...
    // With one workgroup resident per CU nothing covers the hand-over between two workgroups (dispatch, the bias block, the cold ring):
    // here the bias block is loaded once, and the weight stream of the next tile is re-issued behind the last barrier of this one, in
    // front of the tile's output, so that it lands while the next tile's rays are loaded and encoded.
    // (TRAIN: the store is tiled in groups of 8 sample tiles and the backward walks all of them -- the padding tiles are written too)
    const long ntile = TRAIN ? (p.nsamp + 255) / 256 * (256 / CCfg::SAMPLES) : (p.nsamp + CCfg::SAMPLES - 1) / CCfg::SAMPLES;
    for (long tile = blockIdx.x;;) {
    NerfCtxC<N, ST> cx;                        // per tile: nothing of it but the stream's four address words is carried around the loop
    cx.st = st0;
    cx.bias = lbias;
    cx.lane = lane;
    const long smp = tile * CCfg::SAMPLES + wave * 32 + n;
    const bool valid = smp < p.nsamp;
    const long sidx = valid ? smp : p.nsamp - 1;
    float z_own = 0.f, z_next = 0.f, dnorm = 0.f;
    {   // both positional encodings as input blocks
        const long ray = sidx / p.S;
        const float* rb = p.ray_batch + ray * p.ncol;
        float zv;
        if constexpr (FUSE) {
            const int si = (int)(sidx - ray * p.S);
            zv = c_z_sample(rb[6], rb[7], p.S, p.lindisp, p.perturb, p.t_rand, ray, si);
            z_next = si < p.S - 1 ? c_z_sample(rb[6], rb[7], p.S, p.lindisp, p.perturb, p.t_rand, ray, si + 1) : zv;
            z_own = zv;
            dnorm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(rb[3], rb[3]), __fmul_rn(rb[4], rb[4])), __fmul_rn(rb[5], rb[5])));
            if (p.z_out && valid && h == 0) p.z_out[sidx] = zv;
        } else {
            zv = p.z[sidx];
        }
        float pts[3], vd[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            pts[c] = __fadd_rn(rb[c], __fmul_rn(rb[3 + c], zv));   // renderer.py:180
            vd[c] = rb[8 + c];
        }
        c_encode<PE_L, PE_KS>(pts, h, cx.pe);
        c_encode<PE_LV, PEV_KS>(vd, h, cx.pev);
    }
    cx.act = CAct{};
    if constexpr (TRAIN) {
        const long t32 = __builtin_amdgcn_readfirstlane((int)(tile * (CCfg::SAMPLES / 32) + wave));      // this wavefront's 32-sample tile
        cx.act.base = p.act + t32 * astore::TILE_BYTES;
        cx.act.voff = lane * 16;
        c_store_input<PE_KS>(cx.act, astore::PE, cx.pe);
        c_store_input<PEV_KS>(cx.act, astore::DIR, cx.pev);
    }

    cx.st.start_wait();
    c_prime<typename N::L0>(cx.st, cx.pp, lbias, lane);
    {
        XBlk in_pe[1];
        in_pe[0] = cx.pe;
        c_layer<typename N::L0, typename N::template Hidden<1>, ST, 1, KB, TRAIN>(cx.st, cx.pp, in_pe, cx.buf[0], nullptr, lbias, lane, cx.act);
    }
    HiddenLoopC<N, ST, 1, D, TRAIN>::run(cx);

    // heads
    XBlk (&hact)[KB] = cx.buf[(D - 1) & 1];
    lds_f32_p lb = lbias + D * T * 32;
    float araw[4], rraw[4];
    XBlk none[1];
    c_layer<typename N::Alpha, typename N::Feature, ST, KB, 1, TRAIN>(cx.st, cx.pp, hact, none, araw, lb, lane, cx.act);
    lb += 32;
    XBlk vin[KB + 1];
    c_layer<typename N::Feature, typename N::Views, ST, KB, KB + 1, TRAIN>(cx.st, cx.pp, hact, vin, nullptr, lb, lane, cx.act);
    lb += T * 32;
    vin[KB] = cx.pev;
    XBlk hbuf[KB / 2];
    c_layer<typename N::Views, typename N::Rgb, ST, KB + 1, KB / 2, TRAIN>(cx.st, cx.pp, vin, hbuf, nullptr, lb, lane, cx.act);
    lb += (T / 2) * 32;
    c_layer<typename N::Rgb, void, ST, KB / 2, 1, TRAIN>(cx.st, cx.pp, hbuf, none, rraw, lb, lane, cx.act);
    const bool more = tile + gridDim.x < ntile;
    if (more) cx.st.restart_issue();            // every wavefront is behind the barrier that ended the stream's last chunk: all slots are free

    if (h == 0 && valid && (!FUSE || p.raw)) {
        f32x4 o = {rraw[0], rraw[1], rraw[2], araw[0]};   // cat([rgb, alpha]) nerf.py:157
#ifdef EVD_C_STAMP      // lane 0 of every wavefront reports (cycles in the vmcnt waits, in the barriers, in the kernel) instead of its sample
        if (lane == 0) o = f32x4{(float)cx.st.tw, (float)cx.st.tb, (float)(__builtin_readcyclecounter() - t_start), -1.f};
#endif
        *reinterpret_cast<f32x4*>(p.raw + sidx * 4) = o;
    }
    if constexpr (FUSE) {
        // raw2outputs (nerf.py:74-129; the per-sample arithmetic of k_composite_rows) on the values in registers
        const long ray = sidx / p.S;
        const int si = (int)(sidx - ray * p.S);
        const bool live = h == 0 && valid;                      // lanes 32..63 mirror the samples: neutral elements
        float alpha = 0.f;
        if (live) {
            if (si < p.S - 1) {
                const float dist = __fmul_rn(__fsub_rn(z_next, z_own), dnorm);
                alpha = __fadd_rn(-expf(-__fmul_rn(act_fast(p.sigma_act, araw[0]), dist)), 1.f);
            } else {
                alpha = 1.f;                                    // last sample: alpha forced to 1 (nerf.py:113-114)
            }
        }
        const float om = live ? __fadd_rn(-alpha, 1.f) : 1.f;
        const float incl = wave_scan_mul_dpp(om);
        float T = dpp_f32<0x138>(1.f, incl);                    // wave_shr:1 -> exclusive product; lane 0 keeps 1
        float* xw = reinterpret_cast<float*>(smem + CCfg::RING) + CCfg::BIAS_WORDS - 64;    // 64 spare words behind the row scales
        const int wpr = p.S / 32, wr = wave % wpr, w0 = wave - wr;                           // wavefronts per ray, this one's rank
        if (lane == 31) xw[wave] = incl;                        // the wavefront's total
        __syncthreads();
        float Tb = 1.f;
        for (int k = 0; k < wr; ++k) Tb *= xw[w0 + k];          // the transmittance the ray arrives with
        const float wgt = live ? alpha * (Tb * T) : 0.f;
        if (p.weights && live) p.weights[sidx] = wgt;
        float part[5] = {wgt, wgt * z_own, wgt * act_fast(p.rgb_act, rraw[0]), wgt * act_fast(p.rgb_act, rraw[1]), wgt * act_fast(p.rgb_act, rraw[2])};
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            part[q] = wave_sum_dpp(part[q]);
            if (lane == 0) xw[8 + wave * 5 + q] = part[q];
        }
        __syncthreads();
        if (wr == 0 && lane == 0 && valid) {
            float tot[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < wpr; ++k)
#pragma unroll
                for (int q = 0; q < 5; ++q) tot[q] += xw[8 + (w0 + k) * 5 + q];
            if (p.acc_map) p.acc_map[ray] = tot[0];
            if (p.depth_map) p.depth_map[ray] = tot[1];
            if (p.rgb_map) {
                const float bg = p.white_bkgd ? 1.f - tot[0] : 0.f;
#pragma unroll
                for (int c = 0; c < 3; ++c) p.rgb_map[ray * 3 + c] = tot[2 + c] + bg;
            }
        }
    }
#ifdef EVD_C_STAMP
    st0.tw = cx.st.tw; st0.tb = cx.st.tb;
#endif
    if (!more) break;
    tile += gridDim.x;
    }
}

template <int W, int D, int SKIP> constexpr int nerf_c_chunks() { return NerfNetC<W, D, SKIP>::NCH; }

template <int W, int D, int SKIP, bool FUSE, bool TRAIN = false>
static int launch_nerf_c(const MlpParams& p, hipStream_t st) {
    typedef NerfNetC<W, D, SKIP, TRAIN> N;
    const long tiles = TRAIN ? cdiv(p.nsamp, 256L) * (256 / CCfg::SAMPLES) : cdiv(p.nsamp, CCfg::SAMPLES);
    const long blocks = cmin_l(tiles, (long)c_persistent_blocks());
    const size_t lds = CCfg::TOTAL;
    EVD_SET_MAX_LDS((&k_nerf_mlp_c<W, D, SKIP, FUSE, TRAIN>), lds);
    if (TRAIN != (p.act != nullptr)) return fail(EVD_E_INVALID, "evd_nerf_mlp (f16c): the activation store goes with the training launch");
    if (p.nbias != N::NTILES * 32) return fail(EVD_E_INVALID, "evd_nerf_mlp (f16c): %d bias floats, kernel expects %d", p.nbias, N::NTILES * 32);
    if (p.nchunks != N::NCH) return fail(EVD_E_INVALID, "evd_nerf_mlp (f16c): packed stream has %d chunks, kernel expects %d", p.nchunks, N::NCH);
    if (!p.wscale) return fail(EVD_E_INVALID, "evd_nerf_mlp (f16c): no row scales");
    if (FUSE && !(p.S == 32 || p.S == 64 || p.S == 128)) return fail(EVD_E_INVALID, "evd_nerf_mlp (f16c, fused step): N_samples must be 32, 64 or 128");
    hipLaunchKernelGGL((k_nerf_mlp_c<W, D, SKIP, FUSE, TRAIN>), dim3((unsigned)blocks), dim3(CCfg::NT), lds, st, p);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

constexpr bool nerf_c_built(int W, int D, int skip) { return W == 256 && D == 8 && skip == 4; }


}  // namespace evd
