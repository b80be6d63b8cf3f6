// Generic fused NeRF backbone (any depth / skip index, all four arithmetic modes; the float32 and split-float16
// modes always run here) + the dispatch to the software-pipelined kernels of nerf_mlp_kernel.h where they are built.
// pts = o + d z -> positional encodings -> D x W MLP (skip + view branch) -> raw.
// reference: networks/nerf.py:46-72 (mlpforward), :131-162 (eval), networks/embedding.py:88-98, renderer.py:180.
//
// Design (see nerf_mlp.h for the fragment/permutation contract):
//   * one wavefront owns 32 samples for the whole network; activations never leave registers
//     (the MFMA D fragment of a layer IS the B fragment of the next one),
//   * weights are the A operand: one pre-permuted stream of MFMA fragments per precision, staged
//     global -> registers -> LDS in double-buffered chunks shared by all wavefronts of the workgroup
//     (one barrier per chunk), read back conflict-free with ds_read_b128 (lane-linear fragments),
//   * three arithmetic modes share the code: bf16 (32x32x16), split-float16 x3 (32x32x16, hi/lo operands,
//     2^11-scaled cross terms in a second accumulator) and exact float32 (32x32x2).
#include "nerf_mlp_kernel.h"

namespace evd {

// VKS: k-steps of the direction encoding (2: multires_views <= 4, 4: <= 10)
template <int PREC, int W, int VKS>
__global__ __launch_bounds__(mlp_threads(PREC), is_half_prec(PREC) ? 2 : 1) void k_nerf_mlp_generic(const MlpParams p) {
    typedef Ops<PREC> O;
    typedef typename O::B B;
    constexpr int NT = mlp_threads(PREC);
    constexpr int T = W / 32, KS = W / 16;
    constexpr int FPC = Stream<PREC>::FPC;
    static_assert(W % 64 == 0, "W must be a multiple of 64");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, h = lane >> 5;
    const long s = (long)blockIdx.x * (NT / 2) + wave * 32 + n;
    const bool valid = s < p.nsamp;
    const long sc = valid ? s : p.nsamp - 1;
    const long ray = sc / p.S;
    const float* rb = p.ray_batch + ray * p.ncol;
    const float zv = p.z[sc];
    float pts[3], vd[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        pts[c] = __fadd_rn(rb[c], __fmul_rn(rb[3 + c], zv));   // renderer.py:180
        vd[c] = p.no_views ? 0.f : rb[8 + c];
    }

    // point encoding straight into B-fragment order (nerf_mlp.h); a copy is parked in this wavefront's LDS stash
    // for the skip layer instead of occupying registers through layers 1..skip
    float* bias = stage_bias<PREC>(smem, p.bias, p.nbias, tid);
    B* stash = reinterpret_cast<B*>(smem + MlpLds<PREC>::RING + MlpLds<PREC>::BIAS_FLOATS * 4 + wave * MlpLds<PREC>::STASH_PER_WAVE) + lane;
    B act[KS], nxt[KS];
    {
        B in_pe[PE_KS];
        encode_b_rt<PREC, PE_KS>(pts, h, p.pe_l, in_pe);
#pragma unroll
        for (int j = 0; j < PE_KS; ++j) stash[j * 64] = in_pe[j];
    }
    Stream<PREC> st;
    st.start(p.wstream, smem, p.nchunks, tid);
    float* frow = (p.feature && valid) ? p.feature + s * W : nullptr;
    {
        B in_pe[PE_KS];
#pragma unroll
        for (int j = 0; j < PE_KS; ++j) in_pe[j] = stash[j * 64];
        layer<PREC, PE_KS, T, true, OUT_B, 0, true>(st, in_pe, nxt, nullptr, bias, lane, (p.feature_kind == 2 && p.D == 1) ? frow : nullptr, W);
    }
    bias += T * 32;
#pragma unroll
    for (int j = 0; j < KS; ++j) act[j] = nxt[j];

#pragma nounroll
    for (int l = 1; l < p.D; ++l) {
        float* fr = (p.feature_kind == 2 && l == p.D - 1) ? frow : nullptr;
        if (l - 1 == p.skip) {      // h = cat([input_pts, h]) feeds this layer (nerf.py:137-138)
            B wide[PE_KS + KS];
#pragma unroll
            for (int j = 0; j < PE_KS; ++j) wide[j] = stash[j * 64];
#pragma unroll
            for (int j = 0; j < KS; ++j) wide[PE_KS + j] = act[j];
            layer<PREC, PE_KS + KS, T, true, OUT_B, 0, true>(st, wide, nxt, nullptr, bias, lane, fr, W);
        } else {
            layer<PREC, KS, T, true, OUT_B, 0, true>(st, act, nxt, nullptr, bias, lane, fr, W);
        }
        bias += T * 32;
#pragma unroll
        for (int j = 0; j < KS; ++j) act[j] = nxt[j];
    }

    if (p.no_views) {          // use_viewdirs=False: outputs = output_linear(h) (nerf.py:158-160), rows 0..3 = (rgb, sigma)
        float oraw[16];
        layer<PREC, KS, 1, false, OUT_F32, 0, true>(st, act, nullptr, oraw, bias, lane, nullptr, W);
        if (h == 0 && valid) {
            const f32x4 o = {oraw[0], oraw[1], oraw[2], oraw[3]};
            *reinterpret_cast<f32x4*>(p.raw + s * 4) = o;
        }
        return;
    }
    // heads: alpha_linear (1 tile), feature_linear, views_linears.0 on cat([feature, dirs]), rgb_linear (nerf.py:144-157)
    constexpr int F_ALPHA = KS, F_FEAT = T * KS, F_VIEWS = (T / 2) * (KS + VKS);
    constexpr int OFF_FEAT = F_ALPHA % FPC, OFF_VIEWS = (F_ALPHA + F_FEAT) % FPC, OFF_RGB = (F_ALPHA + F_FEAT + F_VIEWS) % FPC;
    float araw[16], rraw[16];
    layer<PREC, KS, 1, false, OUT_F32, 0, false>(st, act, nullptr, araw, bias, lane, nullptr, W);
    bias += 32;
    layer<PREC, KS, T, false, OUT_B, OFF_FEAT, false>(st, act, nxt, nullptr, bias, lane, p.feature_kind == 1 ? frow : nullptr, W);
    bias += T * 32;
    {
        B vin[KS + VKS];
#pragma unroll
        for (int j = 0; j < KS; ++j) vin[j] = nxt[j];
        {
            B in_dir[VKS];
            encode_b_rt<PREC, VKS>(vd, h, p.pe_lv, in_dir);
#pragma unroll
            for (int j = 0; j < VKS; ++j) vin[KS + j] = in_dir[j];
        }
        layer<PREC, KS + VKS, T / 2, true, OUT_B, OFF_VIEWS, false>(st, vin, act, nullptr, bias, lane, nullptr, W);
        bias += (T / 2) * 32;
    }
    {
        B hin[KS / 2];
#pragma unroll
        for (int j = 0; j < KS / 2; ++j) hin[j] = act[j];
        layer<PREC, KS / 2, 1, false, OUT_F32, OFF_RGB, true>(st, hin, nullptr, rraw, bias, lane, nullptr, W);
    }
    if (h == 0 && valid) {
        const f32x4 o = {rraw[0], rraw[1], rraw[2], araw[0]};   // cat([rgb, alpha]) nerf.py:157
        *reinterpret_cast<f32x4*>(p.raw + s * 4) = o;
    }
}

template <int PREC, int W, int VKS>
static int launch_mlp(const MlpParams& p, hipStream_t st) {
    constexpr int NT = mlp_threads(PREC);
    const long blocks = cdiv(p.nsamp, NT / 2);
    const size_t lds = MlpLds<PREC>::TOTAL;
    EVD_SET_MAX_LDS((&k_nerf_mlp_generic<PREC, W, VKS>), lds);
    if (p.nbias > MlpLds<PREC>::BIAS_FLOATS) return fail(EVD_E_INVALID, "evd_nerf_mlp: %d bias floats exceed the LDS bias block", p.nbias);
    hipLaunchKernelGGL((k_nerf_mlp_generic<PREC, W, VKS>), dim3((unsigned)blocks), dim3(NT), lds, st, p);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

int nerf_mlp_generic_dispatch(int prec, int W, const MlpParams& p, hipStream_t st) {
    if (p.pe_l < 0 || p.pe_l > PE_L_MAX || p.pe_lv < 0 || pe_ksteps(p.pe_lv) > 4)
        return fail(EVD_E_INVALID, "evd_nerf_mlp: multires %d / multires_views %d out of range (0..%d / 0..10)", p.pe_l, p.pe_lv, PE_L_MAX);
    const bool widev = pe_ksteps(p.pe_lv) > 2;
#define EVD_CASE(P, WW) if (prec == P && W == WW) return widev ? launch_mlp<P, WW, 4>(p, st) : launch_mlp<P, WW, 2>(p, st)
    EVD_CASE(EVD_PREC_BF16, 256);
    EVD_CASE(EVD_PREC_F16, 256);
    EVD_CASE(EVD_PREC_F16X3, 256);
    EVD_CASE(EVD_PREC_F32, 256);
    EVD_CASE(EVD_PREC_BF16, 64);
    EVD_CASE(EVD_PREC_F16, 64);
    EVD_CASE(EVD_PREC_F16X3, 64);
    EVD_CASE(EVD_PREC_F32, 64);
#undef EVD_CASE
    return fail(EVD_E_INVALID, "evd_nerf_mlp: no kernel for precision %d, width %d (built: W in {64,256})", prec, W);
}

int nerf_mlp_pipe_dispatch(int prec, const MlpParams& p, hipStream_t st) {
    const bool feat = p.feature != nullptr;
    if (prec == EVD_PREC_BF16) return launch_nerf_pipe_bf16(feat, p, st);
    if (prec == EVD_PREC_F16) return launch_nerf_pipe_f16(feat, p, st);
    if (prec == EVD_PREC_F16X3) return launch_nerf_pipe_f16x3(feat, p, st);
    return fail(EVD_E_INVALID, "evd_nerf_mlp: no pipelined kernel for precision %d", prec);
}

}  // namespace evd
