// The WHOLE backward of the AWP sample-feature embedding (128 -> 64 -> 64 -> 64 -> 64, awp_embed.h; reference: autograd of
// AdaptiveWeightProposal.sample_feature_embed_layer, networks/dpnerf/awp.py:36-37,98-100) in ONE launch, on the plan of
// voxel_bwd_fused64.h (round 5): a wavefront owns a 32-sample tile from the d h_local rows to d geo, the gradient between the layers
// stays in registers, the weight gradients of the three 64 x 64 layers are accumulator blocks of the wavefront (3 x 4 blocks of 32 x 32 and
// one shared block for their six bias row sums: 208 registers, one wavefront per SIMD), W^T of the four layers sits in LDS.
// Replaces k_awp_rows_to_frags + 3 x (k_wgrad + k_wgrad_reduce) + 4 x k_dgrad_layer: per tile 8 KiB of rows + 16 KiB of stored fragments
// in, 4 KiB of d e0 and 8 KiB of d geo fragments out; layer 0's wgrad (its 8 blocks over the 128 geo columns did not fit next to the
// others: 207 spilled registers) stays k_wgrad on (d e0, geo): 48 KiB per tile in all instead of ~100 KiB through the per-layer chain.
#pragma once

#include "awp_embed.h"
#include "voxel_bwd_fused64.h"

namespace evd {

namespace awpf {
constexpr int KW = AWP_W / 16, KIN = AWP_IN / 16;
// accumulator blocks [row tile][column tile] per layer, top layer first; then the shared bias block (column 2 (3 - l) + row tile)
constexpr int A_L3 = 0, A_L2 = 4, A_L1 = 8, A_BIAS = 12, NBLK = 13;
constexpr int acc0(int l) { return l == 3 ? A_L3 : l == 2 ? A_L2 : A_L1; }
// W^T fragments in LDS: layers 3, 2, 1 [2 output tiles][4 k-steps], layer 0 [4 output tiles][4 k-steps]
constexpr int W_L3 = 0, W_L2 = 8, W_L1 = 16, W_L0 = 24, W_N = 40;
constexpr int wt0(int l) { return l == 3 ? W_L3 : l == 2 ? W_L2 : l == 1 ? W_L1 : W_L0; }
constexpr int LDS_BYTES = W_N * 1024 + 4 * 4096;
}  // namespace awpf

struct AwpBwdFusedParams {
    const float* d_h_local;             // [nsamp, 64]
    long nsamp, tiles;
    char* store;
    const char* wt[AWP_D];
    unsigned* words;                    // the store's trailer: [0] loss-scale word (in), [1] max |d geo| in true units (out, atomicMax)
    float* partial;                     // [gridDim.x][awpf::NBLK][64 lanes][16] float32
};

template <int PREC>
__global__ __launch_bounds__(256, 1) void k_awp_bwd_fused(const AwpBwdFusedParams p) {
    using namespace awpf;
    using namespace awpstore;
    static_assert(is_half_prec(PREC) && AWP_D == 4 && AWP_W == 64 && AWP_IN == 128, "the shipped embedding");
    typedef POps<PREC> O;
    extern __shared__ __attribute__((aligned(16))) char asm_[];
    char* wl = asm_;
    float* fold = reinterpret_cast<float*>(asm_ + W_N * 1024);
    pipe_fp16_saturate<PREC>();
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), n = lane & 31, h = lane >> 5;
#pragma unroll
    for (int l = 0; l < AWP_D; ++l) {
        const int cnt = (l == 0 ? 16 : 8) * 64;
        for (int i = tid; i < cnt; i += 256) *reinterpret_cast<f32x4*>(wl + wt0(l) * 1024 + i * 16) = *reinterpret_cast<const f32x4*>(p.wt[l] + (long)i * 16);
    }
    __syncthreads();
    auto WT = [&](int f) -> W4 { return *reinterpret_cast<const W4*>(wl + f * 1024 + lane * 16); };
    const unsigned one = half_one_pair<PREC>();
    auto selectors = [&](W4& sel0, W4& sel1) {
        int nn = n, hh = h;
        asm volatile("" : "+v"(nn), "+v"(hh));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int kk = 8 * hh + 2 * e;
            sel0.w[e] = (nn == kk ? (one & 0xffffu) : 0u) | (nn == kk + 1 ? (one & 0xffff0000u) : 0u);
            sel1.w[e] = (nn == 16 + kk ? (one & 0xffffu) : 0u) | (nn == 17 + kk ? (one & 0xffff0000u) : 0u);
        }
    };
    auto ones = [&](int c) {
        int nn = n;
        asm volatile("" : "+v"(nn));
        const unsigned v = nn == c ? one : 0u;
        return W4{{v, v, v, v}};
    };
    f32x16 acc[NBLK];
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float scale = grad_scale(p.words[0], false);
    auto ld = [&](long t, int slot) -> W4 { return *reinterpret_cast<const W4*>(p.store + t * TILE_BYTES + (long)slot * 1024 + lane * 16); };
    auto prod = [&](f32x16& a, const W4 (&yt)[2], const W4 (&xt)[2]) {
        a = mfma_half<PREC>(yt[0], xt[0], a);
        a = mfma_half<PREC>(yt[1], xt[1], a);
    };
    auto bias = [&](int c, const W4 (&yt)[2]) {
        const W4 o = ones(c);
        acc[A_BIAS] = mfma_half<PREC>(yt[0], o, acc[A_BIAS]);
        acc[A_BIAS] = mfma_half<PREC>(yt[1], o, acc[A_BIAS]);
    };
    auto frags = [&](const f32x16& d, W4& o0, W4& o1) {
        typename O::B o2[2];
#pragma unroll
        for (int k = 0; k < 8; ++k) O::template set_pair<false>(o2[k >> 2], k & 3, d[2 * k], d[2 * k + 1]);
        o0 = __builtin_bit_cast(W4, o2[0]);
        o1 = __builtin_bit_cast(W4, o2[1]);
    };
    auto mask = [&](W4& g, const W4& a) {
#pragma unroll
        for (int e = 0; e < 4; ++e) g.w[e] = mask_word(g.w[e], a.w[e]);
    };
    // one 64 -> 64 layer l (3, 2, 1): wgrad of (gradient g, input activations x), dgrad to the input's gradient, masked by x
    auto layer = [&](int l, const W4 (&g)[4], const W4 (&x)[4], W4 (&gout)[4]) {
        W4 yt[2][2], xt[2], sel0, sel1;
        selectors(sel0, sel1);
#pragma unroll
        for (int yb = 0; yb < 2; ++yb) transpose_block<PREC>(g[2 * yb], g[2 * yb + 1], sel0, sel1, yt[yb]);
#pragma unroll
        for (int xb = 0; xb < 2; ++xb) {
            transpose_block<PREC>(x[2 * xb], x[2 * xb + 1], sel0, sel1, xt);
#pragma unroll
            for (int yb = 0; yb < 2; ++yb) prod(acc[acc0(l) + 2 * yb + xb], yt[yb], xt);
        }
#pragma unroll
        for (int yb = 0; yb < 2; ++yb) bias(2 * (3 - l) + yb, yt[yb]);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            f32x16 d = zero16;
#pragma unroll
            for (int j = 0; j < 4; ++j) d = mfma_half<PREC>(WT(wt0(l) + 4 * r + j), g[j], d);
            frags(d, gout[2 * r], gout[2 * r + 1]);
            mask(gout[2 * r], x[2 * r]);
            mask(gout[2 * r + 1], x[2 * r + 1]);
        }
    };

    const long nw = (long)gridDim.x * 4;
    long t = (long)blockIdx.x * 4 + wave;
    const long last = p.tiles - 1;
    // the tile's inputs by consuming phase, loaded two phases ahead: rows + e3 (top), e2, e1, e0, geo
    f32x4 rlo[4], rhi[4];
    W4 e3[4], e2[4], e1[4], e0[4];
    auto load_top = [&](long tt) {
        tt = tt < last ? tt : last;
        const long smp = tt * 32 + n, sc = smp < p.nsamp ? smp : p.nsamp - 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* r = p.d_h_local + sc * AWP_W + 16 * j + 4 * h;      // fragment j, position 8 h + e <-> channel 16 j + phi(8 h + e): two runs of four
            rlo[j] = *reinterpret_cast<const f32x4*>(r);
            rhi[j] = *reinterpret_cast<const f32x4*>(r + 8);
            e3[j] = ld(tt, E0 + 3 * KW + j);
        }
    };
    auto load4 = [&](W4 (&dst)[4], long tt, int slot) {
        tt = tt < last ? tt : last;
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[j] = ld(tt, slot + j);
    };
    float gmax = 0.f;
    if (t < p.tiles) {
        load_top(t);
        load4(e2, t, E0 + 2 * KW);
        load4(e1, t, E0 + KW);
    }
    for (; t < p.tiles; t += nw) {
        // ---- top: d h_local rows x loss scale -> gradient fragments of layer 3, masked by its ReLU pattern ------------------------------
        load4(e0, t, E0);
        asm volatile("" ::: "memory");
        W4 g3[4], g2[4], g1[4], g0[4];
        {
            const bool in = t * 32 + n < p.nsamp;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 lo = in ? rlo[j] : f32x4{0.f, 0.f, 0.f, 0.f}, hi = in ? rhi[j] : f32x4{0.f, 0.f, 0.f, 0.f};
                typename O::B b;
                O::template set_pair<false>(b, 0, lo[0] * scale, lo[1] * scale);
                O::template set_pair<false>(b, 1, lo[2] * scale, lo[3] * scale);
                O::template set_pair<false>(b, 2, hi[0] * scale, hi[1] * scale);
                O::template set_pair<false>(b, 3, hi[2] * scale, hi[3] * scale);
                g3[j] = __builtin_bit_cast(W4, b);
                mask(g3[j], e3[j]);
            }
        }
        layer(3, g3, e2, g2);
        if (t + nw < p.tiles) load_top(t + nw);
        asm volatile("" ::: "memory");
        layer(2, g2, e1, g1);
        if (t + nw < p.tiles) load4(e2, t + nw, E0 + 2 * KW);
        asm volatile("" ::: "memory");
        layer(1, g1, e0, g0);
        // ---- layer 0: d e0 leaves for its wgrad; d geo = W_0^T d e0 (no activation below the geo features) ------------------------------
        if (t + nw < p.tiles) load4(e1, t + nw, E0 + KW);
        asm volatile("" ::: "memory");
        {
            char* out = p.store + t * TILE_BYTES + lane * 16;
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<W4*>(out + (long)(D_E0 + j) * 1024) = g0[j];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f32x16 d = zero16;
#pragma unroll
                for (int j = 0; j < 4; ++j) d = mfma_half<PREC>(WT(W_L0 + 4 * r + j), g0[j], d);
                W4 o0, o1;
                frags(d, o0, o1);
                *reinterpret_cast<W4*>(out + (long)(D_GEO + 2 * r) * 1024) = o0;
                *reinterpret_cast<W4*>(out + (long)(D_GEO + 2 * r + 1) * 1024) = o1;
                float nan_probe = 0.f;               // fmaxf drops a NaN (and inf - inf): record it as +inf, like k_mam_local_bwd does for |d h_local|
#pragma unroll
                for (int i = 0; i < 16; ++i) { gmax = fmaxf(gmax, fabsf(d[i])); nan_probe += d[i]; }       // (the fragments hold these values rounded to half precision)
                gmax = nan_probe != nan_probe ? __builtin_huge_valf() : gmax;
            }
        }
    }
    // max |d geo| in true units for the fine level's rescaling (k_dgrad_layer absmax_out): one guarded atomicMax per wavefront
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, o));
    if (lane == 0 && gmax > 0.f) {
        const unsigned mb = __float_as_uint(gmax * grad_scale(p.words[0], true));
        if (mb > *reinterpret_cast<volatile unsigned*>(p.words + 1)) atomicMax(p.words + 1, mb);
    }
    float* part = p.partial + (long)blockIdx.x * NBLK * 1024;
#pragma unroll
    for (int b = 0; b < NBLK; ++b) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = {acc[b][4 * q], acc[b][4 * q + 1], acc[b][4 * q + 2], acc[b][4 * q + 3]};
            *reinterpret_cast<f32x4*>(fold + wave * 1024 + lane * 16 + 4 * q) = v;
        }
        __syncthreads();
        f32x4 s = *reinterpret_cast<const f32x4*>(fold + wave * 256 + lane * 4);
#pragma unroll
        for (int w = 1; w < 4; ++w) s += *reinterpret_cast<const f32x4*>(fold + w * 1024 + wave * 256 + lane * 4);
        *reinterpret_cast<f32x4*>(part + b * 1024 + wave * 256 + lane * 4) = s;
        __syncthreads();
    }
}

}  // namespace evd
