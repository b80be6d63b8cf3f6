// Loss-side pixel ops: sub-exposure weighted sums, camera response functions, fused blur / event loss
// reductions, EDI prior.  All HBM-bound and tiny; the point of fusing them is one launch + one packed
// partial-sum vector per step (which is what the ranks all-reduce) instead of dozens of ATen launches.
#include <type_traits>

#include "evd_common.h"

namespace evd {

// CRF parameters travel as a kernel argument (scalar loads, wave-uniform).  networks/tonemapping.py:16-22
constexpr int CRF_MAX_IN = 8;
struct CrfParams {
    int map_type;       // 0 none, 1 gamma, 2 learn
    int E;              // extra features
    float inv_gamma;
    float b3;
    float w0[16 * CRF_MAX_IN], b0[16], w1[256], b1[16], w2[256], b2[16], w3[16];
};

// CRF.forward for one channel value, networks/tonemapping.py:59-93
__device__ __forceinline__ float crf_apply(const CrfParams& c, float v, const float* feat, bool skip_learn) {
    if (c.map_type == 0) return v;
    if (c.map_type == 1) v = powf(v, c.inv_gamma);
    if (!skip_learn && c.map_type == 2) {
        // every index into the parameter block is a compile-time constant (w0 rows are padded to CRF_MAX_IN): the weights
        // stay scalar loads from the kernel-argument buffer.  (A runtime-strided w0[j * nin + k] made hipcc copy the whole
        // 2.4 KB block to scratch, per lane.)
        float in[CRF_MAX_IN];
        in[0] = v;
#pragma unroll
        for (int e = 1; e < CRF_MAX_IN; ++e) in[e] = (feat && e - 1 < c.E) ? feat[e - 1] : 0.f;
        float h[16], h2[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float s = c.b0[j];
#pragma unroll
            for (int k = 0; k < CRF_MAX_IN; ++k) s = fmaf(c.w0[j * CRF_MAX_IN + k], in[k], s);   // padded columns are zero
            h[j] = fmaxf(s, 0.f);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float s = c.b1[j];
#pragma unroll
            for (int k = 0; k < 16; ++k) s = fmaf(c.w1[j * 16 + k], h[k], s);
            h2[j] = fmaxf(s, 0.f);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float s = c.b2[j];
#pragma unroll
            for (int k = 0; k < 16; ++k) s = fmaf(c.w2[j * 16 + k], h2[k], s);
            h[j] = fmaxf(s, 0.f);
        }
        float s = c.b3;
#pragma unroll
        for (int k = 0; k < 16; ++k) s = fmaf(c.w3[k], h[k], s);
        v = 1.f / (1.f + expf(-(s * 0.1f + v)));
    }
    return v;
}

__device__ __forceinline__ float luma_of(int standard, float r, float g, float b) {
    if (standard == 0) return 0.299f * r + 0.587f * g + 0.114f * b;        // rec601, tonemapping.py:128-129
    if (standard == 1) return 0.2126f * r + 0.7152f * g + 0.0722f * b;     // rec709
    return (r + g + b) / 3.f;                                              // avg
}

__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < nw; ++i) s += red[i];
    return s;
}

// rbk_weighted_sum, networks/dpnerf/blurmodel.py:112-127
__global__ void k_weighted_sum(const float* __restrict__ x, const float* __restrict__ ccw, long R, int P, int C,
                               float* __restrict__ out) {
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (idx >= R * C) return;
    const long r = idx / C;
    const int c = idx % C;
    float s = 0.f;
    for (int p = 0; p < P; ++p) s += x[(r * P + p) * (long)C + c] * ccw[r * P + p];
    out[idx] = s;
}

__global__ void k_crf_forward(const CrfParams crf, const float* __restrict__ x, const float* __restrict__ feat,
                              int feat_per_channel, int skip_learn, int luma, long n, float* __restrict__ out) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* f = feat ? (feat_per_channel ? feat + (i * 3 + c) * crf.E : feat + i * crf.E) : nullptr;
        v[c] = crf_apply(crf, x[i * 3 + c], f, skip_learn);
    }
    if (luma < 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) out[i * 3 + c] = v[c];
    } else {
        out[i] = luma_of(luma, v[0], v[1], v[2]);
    }
}

// spec: run_nerf.py:443-497 + networks/renderer.py:327-336
__global__ __launch_bounds__(256) void k_blur_loss(const CrfParams crf, int skip_learn, const float* __restrict__ rgb_p,
                                                   const float* __restrict__ rgb0_p, const float* __restrict__ w1,
                                                   const float* __restrict__ w2, const float* __restrict__ tgt,
                                                   const float* __restrict__ tgt0, long R, int P, float* __restrict__ partial,
                                                   float* __restrict__ o_rgb, float* __restrict__ o_rgb1, float* __restrict__ o_awp) {
    // one lane per (pixel, colour channel): a 1024-pixel blur batch is 3072 lanes in 48 small blocks instead of 4 busy ones
    __shared__ float red[8];
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const long r = idx / 3;
    const int ch = (int)(idx % 3);
    float se[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    float cnt = 0.f;
    if (r < R) {
        cnt = 1.f;
        float a = 0.f, b = 0.f, c = 0.f;
        for (int p = 0; p < P; ++p) {
            const float wa = w1[r * P + p], wb = w2 ? w2[r * P + p] : 0.f;
            const float f = rgb_p[(r * P + p) * 3 + ch];
            a += f * wa;
            c += f * wb;
            if (rgb0_p) b += rgb0_p[(r * P + p) * 3 + ch] * wa;
        }
        const float t = tgt[r * 3 + ch];
        float d = crf_apply(crf, a, nullptr, skip_learn) - t;
        se[0] = d * d;
        if (o_rgb) o_rgb[r * 3 + ch] = a;
        if (rgb0_p) {
            d = crf_apply(crf, b, nullptr, skip_learn) - t;
            se[1] = d * d;
            if (o_rgb1) o_rgb1[r * 3 + ch] = b;
        }
        if (w2) {
            d = crf_apply(crf, c, nullptr, skip_learn) - t;
            se[2] = d * d;
            if (o_awp) o_awp[r * 3 + ch] = c;
        }
        if (tgt0) {
            const float t0 = tgt0[r * 3 + ch];
            d = crf_apply(crf, rgb_p[(r * P) * 3 + ch], nullptr, skip_learn) - t0;     // rgb_pts[:, 0] renderer.py:374
            se[3] = d * d;
            if (rgb0_p) {
                d = crf_apply(crf, rgb0_p[(r * P) * 3 + ch], nullptr, skip_learn) - t0;
                se[4] = d * d;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const float s = block_sum(se[k], red);
        if (threadIdx.x == 0 && s != 0.f) atomicAdd(partial + k, s);
    }
    const float n = block_sum(cnt, red);
    if (threadIdx.x == 0) atomicAdd(partial + 5, n);
}

// d crf(x) / dx for the non-learnable response curves (identity, gamma): tonemapping.py:64-68
__device__ __forceinline__ float crf_grad_simple(const CrfParams& c, float x) {
    if (c.map_type == 1) return c.inv_gamma * powf(x, c.inv_gamma - 1.f);
    return 1.f;
}

// Backward of k_blur_loss ("next" row f-1, slice): with g[k] = dL/d partial[k] (k = 0..4) and a = sum_p w1 rgb_p etc.
//   d a = 2 g0 (crf(a) - t) crf'(a),  d b = 2 g1 (..rgb0..),  d c = 2 g2 (..w2..),  pts0 terms act on p = 0;
//   d rgb_p[p] = d a w1[p] + d c w2[p],   d rgb0_p[p] = d b w1[p],   d w1[p] = d a . rgb_p[p] + d b . rgb0_p[p],   d w2[p] = d c . rgb_p[p].
// One lane per (pixel, channel); d w1 / d w2 sum the three channels of a pixel with two DPP adds inside the lane triple... the
// triples straddle wavefront rows, so the channel sum goes through atomicAdd on the zero-initialised outputs instead.
__global__ __launch_bounds__(64) void k_blur_loss_bwd(const CrfParams crf, int skip_learn, const float* __restrict__ rgb_p,
                                                      const float* __restrict__ rgb0_p, const float* __restrict__ w1,
                                                      const float* __restrict__ w2, const float* __restrict__ tgt,
                                                      const float* __restrict__ tgt0, long R, int P, float g0, float g1, float g2,
                                                      float g3, float g4, const float* __restrict__ gdev, float* __restrict__ d_rgb_p,
                                                      float* __restrict__ d_rgb0_p, float* __restrict__ d_w1, float* __restrict__ d_w2) {
    if (gdev) { g0 = gdev[0]; g1 = gdev[1]; g2 = gdev[2]; g3 = gdev[3]; g4 = gdev[4]; }      // dL/d partial from device memory (no host copy)
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const long r = idx / 3;
    const int ch = (int)(idx % 3);
    if (r >= R) return;
    float a = 0.f, b = 0.f, c = 0.f;
    for (int p = 0; p < P; ++p) {
        const float wa = w1[r * P + p], wb = w2 ? w2[r * P + p] : 0.f;
        const float f = rgb_p[(r * P + p) * 3 + ch];
        a += f * wa;
        c += f * wb;
        if (rgb0_p) b += rgb0_p[(r * P + p) * 3 + ch] * wa;
    }
    const float t = tgt[r * 3 + ch];
    const float da = 2.f * g0 * (crf_apply(crf, a, nullptr, skip_learn) - t) * crf_grad_simple(crf, a);
    const float db = rgb0_p ? 2.f * g1 * (crf_apply(crf, b, nullptr, skip_learn) - t) * crf_grad_simple(crf, b) : 0.f;
    const float dc = w2 ? 2.f * g2 * (crf_apply(crf, c, nullptr, skip_learn) - t) * crf_grad_simple(crf, c) : 0.f;
    for (int p = 0; p < P; ++p) {
        const long q = r * P + p;
        const float f = rgb_p[q * 3 + ch], f0 = rgb0_p ? rgb0_p[q * 3 + ch] : 0.f;
        float dr = da * w1[q] + (w2 ? dc * w2[q] : 0.f), dr0 = db * w1[q];
        if (p == 0 && tgt0) {                           // pts0 / EDI-prior terms on the p = 0 render (renderer.py:374)
            const float t0 = tgt0[r * 3 + ch];
            dr += 2.f * g3 * (crf_apply(crf, f, nullptr, skip_learn) - t0) * crf_grad_simple(crf, f);
            if (rgb0_p) dr0 += 2.f * g4 * (crf_apply(crf, f0, nullptr, skip_learn) - t0) * crf_grad_simple(crf, f0);
        }
        d_rgb_p[q * 3 + ch] = dr;
        if (d_rgb0_p) d_rgb0_p[q * 3 + ch] = dr0;
        if (d_w1) atomicAdd(d_w1 + q, da * f + db * f0);
        if (d_w2 && w2) atomicAdd(d_w2 + q, dc * f);
    }
}

// spec: run_nerf.py:518-570 + utils/events.py:260-284.  The learnable event-CRF (a 1+E -> 16 -> 16 -> 16 -> 1 MLP per
// colour value, ~650 FMAs) is evaluated 12 times per event (start/end x fine/coarse x 3 channels): one LANE per
// evaluation, 16 lanes per event (lane = 4 which + channel, channel 3 idle), the luma / log-difference assembled with
// quad and row DPP moves; weights are wave-uniform scalar loads from the kernel argument.
template <int CTRL>
__device__ __forceinline__ float dppf(float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, src), CTRL, 0xf, 0xf, true));
}

__global__ __launch_bounds__(256) void k_event_loss(const CrfParams crf, int skip_learn, int add_bii_feat, int tonemap_only,
                                                    const float* __restrict__ start, const float* __restrict__ end,
                                                    const float* __restrict__ start0, const float* __restrict__ end0,
                                                    const float* __restrict__ cum_neg, const float* __restrict__ cum_pos,
                                                    float thr_neg, float thr_pos, const unsigned char* __restrict__ cmask,
                                                    float cw0, float cw1, float cw2, int has_cw, long N, float* __restrict__ partial) {
    __shared__ float red[8];
    const int sub = threadIdx.x & 15, which = sub >> 2, c = sub & 3;      // which: 0 end, 1 start, 2 end0, 3 start0
    const long i = blockIdx.x * (long)(blockDim.x >> 4) + (threadIdx.x >> 4);
    const bool on = i < N;
    const long ii = on ? i : N - 1;
    const bool have0 = start0 && end0;
    const float cn = cum_neg[ii], cp = cum_pos[ii];
    int ch = 0;
    if (cmask) for (int k = 0; k < 3; ++k) if (cmask[ii * 3 + k]) ch = k;
    const float* src = which == 0 ? end : which == 1 ? start : which == 2 ? end0 : start0;
    float v = 0.f;
    if (c < 3 && (which < 2 || have0)) {
        float f[2] = {cn, cp};
        const float* fp = nullptr;
        if (add_bii_feat == 1) fp = f;                                            // 'pos-neg' :522-523
        else if (add_bii_feat == 2) { if (c != ch) { f[0] = 0.f; f[1] = 0.f; } fp = f; }   // 'color-pos-neg' :524-531
        v = crf_apply(crf, src[ii * 3 + c], fp, skip_learn);
    }
    // gather the quad's three channel values into every lane of the quad
    const float v0 = dppf<0x00>(v), v1 = dppf<0x55>(v), v2 = dppf<0xaa>(v);      // quad_perm broadcasts of lanes 0, 1, 2
    const float sel[3] = {v0, v1, v2};
    const float lum = tonemap_only ? sel[ch] : luma_of(0, v0, v1, v2);
    const float lg = logf(lum + 1e-5f);
    // pred = log(luma(end)) - log(luma(start)): quads 0 - 1 (fine) and 2 - 3 (coarse) of the 16-lane group
    const float other = dppf<0x104>(lg);                                         // row_shl:4 -> lane l reads lane l + 4
    const float pred = lg - other;
    const float bii = __fadd_rn(__fmul_rn(thr_neg, cn), __fmul_rn(thr_pos, cp));   // run_nerf.py:518-519
    const float cw[3] = {cw0, cw1, cw2};
    const float w = (cmask && has_cw) ? cw[ch] : 1.f;
    const float d = pred - bii;
    float s_f = 0.f, s_c = 0.f, s_w = 0.f;
    if (on && sub == 0) { s_f = d * d * w; s_w = w; }
    if (on && sub == 8 && have0) s_c = d * d * w;
    const float a = block_sum(s_f, red), b = block_sum(s_c, red), cc = block_sum(s_w, red);
    if (threadIdx.x == 0) {
        atomicAdd(partial + 0, a);
        atomicAdd(partial + 1, b);
        atomicAdd(partial + 2, cc);
    }
}

// ------------------------------------------------------------------------------------------------
// Backward of k_event_loss ("next" row f-1, slice): gradients of  g_f * partial[0] + g_c * partial[1]  w.r.t. the four colour
// inputs and w.r.t. the parameters of the learnable event-CRF.  Same lane decomposition as the forward kernel (16 lanes per
// event: which = 0 end, 1 start, 2 end0, 3 start0; channel); every lane re-runs its CRF evaluation keeping the three hidden
// layers, back-propagates  d out -> d (input, parameters),  the parameter contributions are summed over the wavefront with DPP
// adds, over the block in LDS and added to the global gradient once per block.
// Parameter gradient layout (EVD_CRF_NPARAM floats): w0 [16][CRF_MAX_IN] (rows padded), b0 [16], w1 [16][16], b1, w2 [16][16], b2, w3 [16], b3.
constexpr int CRF_NPARAM = 16 * CRF_MAX_IN + 16 + 256 + 16 + 256 + 16 + 16 + 1;

__device__ __forceinline__ float wave_sum_all(float v) {      // sum over the 64 lanes, valid in lane 63's row ends; returned uniform
    v += dppf<0xb1>(v);
    v += dppf<0x4e>(v);
    v += dppf<0x141>(v);
    v += dppf<0x140>(v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
}

__global__ __launch_bounds__(256) void k_event_loss_bwd(const CrfParams crf, int skip_learn, int add_bii_feat, int tonemap_only,
                                                        const float* __restrict__ start, const float* __restrict__ end,
                                                        const float* __restrict__ start0, const float* __restrict__ end0,
                                                        const float* __restrict__ cum_neg, const float* __restrict__ cum_pos,
                                                        float thr_neg, float thr_pos, const unsigned char* __restrict__ cmask,
                                                        float cw0, float cw1, float cw2, int has_cw, long N, float g_f, float g_c,
                                                        const float* __restrict__ gdev, float* __restrict__ d_start, float* __restrict__ d_end,
                                                        float* __restrict__ d_start0, float* __restrict__ d_end0,
                                                        float* __restrict__ d_params) {
    __shared__ float pacc[CRF_NPARAM];
    if (gdev) { g_f = gdev[0]; g_c = gdev[1]; }
    for (int i = threadIdx.x; i < CRF_NPARAM; i += blockDim.x) pacc[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, sub = threadIdx.x & 15, which = sub >> 2, c = sub & 3;
    const long i = blockIdx.x * (long)(blockDim.x >> 4) + (threadIdx.x >> 4);
    const bool on = i < N;
    const long ii = on ? i : N - 1;
    const bool have0 = start0 && end0;
    const bool active = on && c < 3 && (which < 2 || have0);
    const bool learn = crf.map_type == 2 && !skip_learn;
    const float cn = cum_neg[ii], cp = cum_pos[ii];
    int ch = 0;
    if (cmask) for (int k = 0; k < 3; ++k) if (cmask[ii * 3 + k]) ch = k;
    const float* src = which == 0 ? end : which == 1 ? start : which == 2 ? end0 : start0;
    // ---- forward of this lane's evaluation, hidden layers kept
    float in[CRF_MAX_IN], h1[16], h2[16], h3[16];
#pragma unroll
    for (int e = 0; e < CRF_MAX_IN; ++e) in[e] = 0.f;
    float x = 0.f, v = 0.f, xg = 0.f, dgamma = 1.f;
    if (c < 3 && (which < 2 || have0)) {
        x = src[ii * 3 + c];
        xg = x;
        if (crf.map_type == 1) { xg = powf(x, crf.inv_gamma); dgamma = crf.inv_gamma * powf(x, crf.inv_gamma - 1.f); }
        v = xg;
        if (learn) {
            in[0] = xg;
            if (add_bii_feat == 1 || (add_bii_feat == 2 && c == ch)) { in[1] = cn; in[2] = cp; }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                float s = crf.b0[j];
#pragma unroll
                for (int k = 0; k < CRF_MAX_IN; ++k) s = fmaf(crf.w0[j * CRF_MAX_IN + k], in[k], s);
                h1[j] = fmaxf(s, 0.f);
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                float s = crf.b1[j];
#pragma unroll
                for (int k = 0; k < 16; ++k) s = fmaf(crf.w1[j * 16 + k], h1[k], s);
                h2[j] = fmaxf(s, 0.f);
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                float s = crf.b2[j];
#pragma unroll
                for (int k = 0; k < 16; ++k) s = fmaf(crf.w2[j * 16 + k], h2[k], s);
                h3[j] = fmaxf(s, 0.f);
            }
            float s = crf.b3;
#pragma unroll
            for (int k = 0; k < 16; ++k) s = fmaf(crf.w3[k], h3[k], s);
            v = 1.f / (1.f + expf(-(s * 0.1f + xg)));
        }
    }
    // ---- luma / log-difference chain (as k_event_loss) and its derivative
    const float v0 = dppf<0x00>(v), v1 = dppf<0x55>(v), v2 = dppf<0xaa>(v);
    const float sel[3] = {v0, v1, v2};
    const float lum = tonemap_only ? sel[ch] : luma_of(0, v0, v1, v2);
    const float lg = logf(lum + 1e-5f);
    const float nxt = dppf<0x104>(lg), prv = dppf<0x114>(lg);          // row_shl:4 / row_shr:4: the partner quad's log-luma
    const bool is_end = (which & 1) == 0;
    const float pred = is_end ? lg - nxt : prv - lg;
    const float bii = __fadd_rn(__fmul_rn(thr_neg, cn), __fmul_rn(thr_pos, cp));
    const float cw[3] = {cw0, cw1, cw2};
    const float w = (cmask && has_cw) ? cw[ch] : 1.f;
    const float g = which < 2 ? g_f : g_c;
    const float d_lg = (is_end ? 1.f : -1.f) * 2.f * g * w * (pred - bii);
    const float d_lum = d_lg / (lum + 1e-5f);
    const float coef[3] = {0.299f, 0.587f, 0.114f};
    float d_v = 0.f;
    if (active) d_v = tonemap_only ? (c == ch ? d_lum : 0.f) : d_lum * coef[c];
    // ---- CRF backward
    float d_x = d_v;
    if (learn) {
        const float dz = d_v * v * (1.f - v);          // through the sigmoid of (0.1 s + x)
        const float ds = 0.1f * dz;
        float dh3[16], dh2[16], dh1[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) dh3[k] = h3[k] > 0.f ? ds * crf.w3[k] : 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) a = fmaf(crf.w2[j * 16 + k], dh3[j], a);
            dh2[k] = h2[k] > 0.f ? a : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) a = fmaf(crf.w1[j * 16 + k], dh2[j], a);
            dh1[k] = h1[k] > 0.f ? a : 0.f;
        }
        float din0 = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) din0 = fmaf(crf.w0[j * CRF_MAX_IN], dh1[j], din0);
        d_x = (dz + din0) * dgamma;
        // parameter contributions, summed over the wavefront, then into the block's LDS accumulator
        constexpr int O_B0 = 16 * CRF_MAX_IN, O_W1 = O_B0 + 16, O_B1 = O_W1 + 256, O_W2 = O_B1 + 16, O_B2 = O_W2 + 256, O_W3 = O_B2 + 16, O_B3 = O_W3 + 16;
        auto add = [&](int idx, float val) {
            const float t = wave_sum_all(val);
            if (lane == 0) atomicAdd(&pacc[idx], t);
        };
#pragma unroll
        for (int j = 0; j < 16; ++j) {
#pragma unroll
            for (int k = 0; k < 3; ++k) add(j * CRF_MAX_IN + k, dh1[j] * in[k]);
            add(O_B0 + j, dh1[j]);
            add(O_B1 + j, dh2[j]);
            add(O_B2 + j, dh3[j]);
            add(O_W3 + j, ds * h3[j]);
        }
        add(O_B3, ds);
#pragma unroll 4
        for (int j = 0; j < 16; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) { add(O_W1 + j * 16 + k, dh2[j] * h1[k]); add(O_W2 + j * 16 + k, dh3[j] * h2[k]); }
    } else {
        d_x = d_v * dgamma;
    }
    if (active) {
        float* dst = which == 0 ? d_end : which == 1 ? d_start : which == 2 ? d_end0 : d_start0;
        if (dst) dst[ii * 3 + c] = d_x;
    }
    if (learn && d_params) {
        __syncthreads();
        for (int k = threadIdx.x; k < CRF_NPARAM; k += blockDim.x) if (pacc[k] != 0.f) atomicAdd(d_params + k, pacc[k]);
    }
}

// AdaptiveWeightProposal.feature_integration (networks/dpnerf/awp.py:49-77), the AWP consumer's compositing scan, AS WRITTEN
// in the reference: the cumprod of :69-73 runs along the CHANNEL axis of the previous sample's row,
//   Q[0, c] = 1,  Q[s, c] = prod_{c' <= c} (1 - alpha[s-1, c'] + 1e-10),  out[c] = sum_s alpha[s, c] Q[s, c] feat[s, c].
// HBM-bound (reads N x S x C floats once): a wavefront owns one ray, a lane CPL consecutive channels; per sample one
// inclusive product scan over the lanes (DPP) of the previous row's (1 - alpha); 4 sample rows of loads in flight.
__device__ __forceinline__ float awp_scan_mul(float v) {
    auto dpp = [](float old, float src, auto ctrl, auto rmask) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src),
                                                                     decltype(ctrl)::value, decltype(rmask)::value, 0xf, false));
    };
    typedef std::integral_constant<int, 0xf> All;
    v *= dpp(1.f, v, std::integral_constant<int, 0x111>(), All());
    v *= dpp(1.f, v, std::integral_constant<int, 0x112>(), All());
    v *= dpp(1.f, v, std::integral_constant<int, 0x114>(), All());
    v *= dpp(1.f, v, std::integral_constant<int, 0x118>(), All());
    v *= dpp(1.f, v, std::integral_constant<int, 0x142>(), std::integral_constant<int, 0xa>());
    v *= dpp(1.f, v, std::integral_constant<int, 0x143>(), std::integral_constant<int, 0xc>());
    return v;
}

// e^x for x <= 0 on the hardware exp2 (v_exp_f32, 1 ulp) behind one multiply: relative error <= 1e-6 for |x| <= 16 against ~20 instructions of
// the IEEE expf -- the two scans evaluate one exponential per (sample, channel) and were bound by the instruction count, not by HBM
__device__ __forceinline__ float awp_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

template <int CPL>
__global__ __launch_bounds__(256) void k_awp_integrate(const float* __restrict__ feat, const float* __restrict__ z,
                                                       const float* __restrict__ rays_d, long N, int S, int C, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long n = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
    if (n >= N) return;
    const float* d = rays_d + n * 3;
    const float norm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(d[0], d[0]), __fmul_rn(d[1], d[1])), __fmul_rn(d[2], d[2])));
    const float* fr = feat + n * (long)S * C;
    const float* zz = z + n * (long)S;
    float acc[CPL], Q[CPL];
#pragma unroll
    for (int q = 0; q < CPL; ++q) { acc[q] = 0.f; Q[q] = 1.f; }
    constexpr int UN = CPL == 1 ? 8 : 4;                 // sample rows of loads in flight (a row is 256 bytes per wavefront)
    for (int s0 = 0; s0 < S; s0 += UN) {
        float f[UN][CPL], dist[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int s = min(s0 + u, S - 1);
#pragma unroll
            for (int q = 0; q < CPL; ++q) { const int c = lane * CPL + q; f[u][q] = c < C ? fr[(long)s * C + c] : 0.f; }
            dist[u] = s < S - 1 ? __fmul_rn(__fsub_rn(zz[s + 1], zz[s]), norm) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int s = s0 + u;
            if (s < S) {
                float om[CPL], local = 1.f;
#pragma unroll
                for (int q = 0; q < CPL; ++q) {
                    const int c = lane * CPL + q;
                    const float alpha = s < S - 1 ? __fadd_rn(-awp_exp(-__fmul_rn(f[u][q], dist[u])), 1.f) : 0.f;   // awp.py:66-67
                    acc[q] = __fadd_rn(acc[q], __fmul_rn(__fmul_rn(alpha, Q[q]), f[u][q]));
                    om[q] = c < C ? __fadd_rn(-alpha, 1.f + 1e-10f) : 1.f;
                    local *= om[q];
                }
                // Q of the NEXT sample row: inclusive product over the channels of this row's (1 - alpha)
                const float incl = awp_scan_mul(local);
                float excl = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, 1.f), __builtin_bit_cast(int, incl), 0x138, 0xf, 0xf, false));
#pragma unroll
                for (int q = 0; q < CPL; ++q) { excl *= om[q]; Q[q] = excl; }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < CPL; ++q) { const int c = lane * CPL + q; if (c < C) out[n * C + c] = acc[q]; }
}

// Backward of k_awp_integrate (the autograd node behind AdaptiveWeightProposal.feature_integration under training, awp.py:98-104):
//   out[c] = sum_s a[s,c] Q[s,c] f[s,c],   a = 1 - e, e = exp(-f dist[s]) (a = 0 on the last sample),   om = e + 1e-10,
//   Q[s+1,c] = prod_{c' <= c} om[s,c'] (the reference's cumprod runs along the CHANNEL axis, awp.py:69-73), Q[0,c] = 1.
// With g[c] = d out[c]:
//   d f[s,c]  = g[c] Q[s,c] (a + f dist e)                                   direct
//             - dist e Sfx[s+1,c] / om[s,c],  Sfx[s+1,c] = sum_{c'' >= c} g[c''] a[s+1,c''] f[s+1,c''] Q[s+1,c'']      through Q of the next row
//   d dist[s] = sum_c (g[c] Q[s,c] f - Sfx[s+1,c] / om[s,c]) f e             -> d z (dist = (z[s+1] - z[s]) |d|), d rays_d
// Same decomposition as the forward: a wavefront per ray, CPL consecutive channels per lane; per sample row one inclusive product
// scan (Q of the next row) and one suffix sum over the lanes (DPP); one row of lookahead.
__device__ __forceinline__ float awp_scan_add(float v) {
    auto dpp = [](float src, auto ctrl, auto rmask) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, src), decltype(ctrl)::value, decltype(rmask)::value, 0xf, false));
    };
    typedef std::integral_constant<int, 0xf> All;
    v += dpp(v, std::integral_constant<int, 0x111>(), All());
    v += dpp(v, std::integral_constant<int, 0x112>(), All());
    v += dpp(v, std::integral_constant<int, 0x114>(), All());
    v += dpp(v, std::integral_constant<int, 0x118>(), All());
    v += dpp(v, std::integral_constant<int, 0x142>(), std::integral_constant<int, 0xa>());
    v += dpp(v, std::integral_constant<int, 0x143>(), std::integral_constant<int, 0xc>());
    return v;
}

template <int CPL>
__global__ __launch_bounds__(256) void k_awp_integrate_bwd(const float* __restrict__ feat, const float* __restrict__ z, const float* __restrict__ rays_d,
                                                           const float* __restrict__ d_out, long N, int S, int C, float* __restrict__ d_feat,
                                                           float* __restrict__ d_z, float* __restrict__ d_rays_d) {
    const int lane = threadIdx.x & 63;
    const long n = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
    if (n >= N) return;
    const float* d = rays_d + n * 3;
    const float norm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(d[0], d[0]), __fmul_rn(d[1], d[1])), __fmul_rn(d[2], d[2])));
    const float* fr = feat + n * (long)S * C;
    const float* zz = z + n * (long)S;
    constexpr int PF = CPL == 1 ? 4 : 2;                 // rows per block; the NEXT block's rows are loaded while this one is processed
    float g[CPL], Q[CPL], cur[PF + 1][CPL], nxt[PF][CPL], ec[CPL];
    auto row = [&](int s, float (&dst)[CPL]) {
        const int sc = s < S ? s : S - 1;
#pragma unroll
        for (int q = 0; q < CPL; ++q) { const int c = lane * CPL + q; dst[q] = c < C ? fr[(long)sc * C + c] : 0.f; }
    };
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        const int c = lane * CPL + q;
        g[q] = c < C ? d_out[n * C + c] : 0.f;
        Q[q] = 1.f;
    }
#pragma unroll
    for (int u = 0; u <= PF; ++u) row(u, cur[u]);
    {   // e of row 0 (every row's exponential is evaluated once: as "the next row's" in the iteration before)
        const float dist0 = S > 1 ? __fmul_rn(__fsub_rn(zz[1], zz[0]), norm) : 0.f;
#pragma unroll
        for (int q = 0; q < CPL; ++q) ec[q] = S > 1 ? awp_exp(-__fmul_rn(cur[0][q], dist0)) : 1.f;
    }
    float dnorm = 0.f, dz_prev = 0.f;                    // d z[s] carried from the previous interval (+ d dist[s-1] |d|)
    for (int s0 = 0; s0 < S; s0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) row(s0 + PF + 1 + u, nxt[u]);
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int s = s0 + u;
            if (s < S) {
                const float (&fc)[CPL] = cur[u];
                const float (&fn)[CPL] = cur[u + 1];
                const bool last = s == S - 1;
                const float dz = last ? 0.f : __fsub_rn(zz[s + 1], zz[s]);
                const float dist = __fmul_rn(dz, norm);
                const float dist_n = s + 2 < S ? __fmul_rn(__fsub_rn(zz[s + 2], zz[s + 1]), norm) : 0.f;
                // this row: e, a, om; Q of the next row
                float e[CPL], en[CPL], om[CPL], Qn[CPL], local = 1.f;
#pragma unroll
                for (int q = 0; q < CPL; ++q) {
                    const int c = lane * CPL + q;
                    e[q] = last ? 1.f : ec[q];
                    om[q] = c < C ? __fadd_rn(last ? 1.f : e[q], 1e-10f) : 1.f;      // the last row's alpha is 0
                    local *= om[q];
                }
                const float incl = awp_scan_mul(local);
                float excl = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, 1.f), __builtin_bit_cast(int, incl), 0x138, 0xf, 0xf, false));
#pragma unroll
                for (int q = 0; q < CPL; ++q) { excl *= om[q]; Qn[q] = excl; }
                // suffix sums over the channels of G[s+1, c] = g a f Q of the next row
                float G[CPL], lsum = 0.f;
#pragma unroll
                for (int q = 0; q < CPL; ++q) {
                    en[q] = s + 2 < S ? awp_exp(-__fmul_rn(fn[q], dist_n)) : 1.f;         // e of row s+1 (its alpha is 0 when it is the last row)
                    const float an = s + 2 < S ? __fadd_rn(-en[q], 1.f) : 0.f;
                    G[q] = last ? 0.f : g[q] * an * fn[q] * Qn[q];
                    lsum += G[q];
                }
                const float pre_incl = awp_scan_add(lsum);                                   // sum over lanes <= this one
                const float total = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pre_incl), 63));
                float sfx = total - pre_incl;                                                 // lanes > this one
                float ddist = 0.f;
#pragma unroll
                for (int q = CPL - 1; q >= 0; --q) {
                    const int c = lane * CPL + q;
                    sfx += G[q];                                                              // channels >= c
                    const float a = last ? 0.f : __fadd_rn(-e[q], 1.f);
                    const float through = last ? 0.f : sfx / om[q];
                    const float ga = g[q] * Q[q] * fc[q] - through;                           // d out / d a[s,c]
                    const float df = last ? 0.f : g[q] * Q[q] * a + ga * dist * e[q];
                    if (c < C) d_feat[(n * (long)S + s) * C + c] = df;
                    ddist += last ? 0.f : ga * fc[q] * e[q];
                }
                if (d_z || d_rays_d) {
                    ddist = awp_scan_add(ddist);
                    ddist = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ddist), 63));
                    if (d_z && lane == 0) d_z[n * (long)S + s] = dz_prev - ddist * norm;
                    dz_prev = ddist * norm;
                    dnorm += ddist * dz;
                }
#pragma unroll
                for (int q = 0; q < CPL; ++q) { Q[q] = Qn[q]; ec[q] = en[q]; }
            }
        }
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            cur[0][q] = cur[PF][q];
#pragma unroll
            for (int u = 0; u < PF; ++u) cur[u + 1][q] = nxt[u][q];
        }
    }
    if (d_rays_d && lane < 3) d_rays_d[n * 3 + lane] = norm > 0.f ? dnorm * d[lane] / norm : 0.f;
}

// The same two scans for C = 64 (the AWP embedding's width, every shipped config): 16 lanes per ray with 4 consecutive channels each (one
// 16-byte load per sample row), FOUR rays per wavefront.  The 64-lane form above spends one wavefront instruction per (ray, sample,
// operation) on 64 channels and was bound by its instruction count (2.0 TB/s); here an instruction serves four rays and the cumulative
// product / suffix sum over the channels is a 4-step scan inside a DPP row.
template <int CTRL>
__device__ __forceinline__ float awp_dpp(float old, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float awp_row_scan_mul(float v) {         // inclusive product over lanes <= this one of the 16-lane row
    v *= awp_dpp<0x111>(1.f, v);
    v *= awp_dpp<0x112>(1.f, v);
    v *= awp_dpp<0x114>(1.f, v);
    v *= awp_dpp<0x118>(1.f, v);
    return v;
}
__device__ __forceinline__ float awp_row_scan_add_right(float v) {   // inclusive sum over lanes >= this one of the row
    v += awp_dpp<0x101>(0.f, v);
    v += awp_dpp<0x102>(0.f, v);
    v += awp_dpp<0x104>(0.f, v);
    v += awp_dpp<0x108>(0.f, v);
    return v;
}
__device__ __forceinline__ float awp_row_sum(float v) {              // the row's sum in every lane
    v += awp_dpp<0xb1>(0.f, v);
    v += awp_dpp<0x4e>(0.f, v);
    v += awp_dpp<0x141>(0.f, v);
    v += awp_dpp<0x140>(0.f, v);
    return v;
}

__global__ __launch_bounds__(256) void k_awp_integrate_c64(const float* __restrict__ feat, const float* __restrict__ z,
                                                           const float* __restrict__ rays_d, long N, int S, float* __restrict__ out) {
    const int l16 = threadIdx.x & 15;
    const long n0 = blockIdx.x * 16L + (threadIdx.x >> 4);
    const long n = n0 < N ? n0 : N - 1;                   // (rays past the end compute on the last ray and write nothing: the DPP rows stay whole)
    const float* d = rays_d + n * 3;
    const float norm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(d[0], d[0]), __fmul_rn(d[1], d[1])), __fmul_rn(d[2], d[2])));
    const float4* fr = reinterpret_cast<const float4*>(feat + n * (long)S * 64) + l16;
    const float* zz = z + n * (long)S;
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, Q[4] = {1.f, 1.f, 1.f, 1.f};
    constexpr int UN = 4;
    for (int s0 = 0; s0 < S; s0 += UN) {
        float4 f4[UN];
        float dist[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int s = min(s0 + u, S - 1);
            f4[u] = fr[(long)s * 16];
            dist[u] = s < S - 1 ? __fmul_rn(__fsub_rn(zz[s + 1], zz[s]), norm) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int s = s0 + u;
            if (s < S) {
                const float f[4] = {f4[u].x, f4[u].y, f4[u].z, f4[u].w};
                float om[4], local = 1.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float alpha = s < S - 1 ? __fadd_rn(-awp_exp(-__fmul_rn(f[q], dist[u])), 1.f) : 0.f;   // awp.py:66-67
                    acc[q] = __fadd_rn(acc[q], __fmul_rn(__fmul_rn(alpha, Q[q]), f[q]));
                    om[q] = __fadd_rn(-alpha, 1.f + 1e-10f);
                    local *= om[q];
                }
                float excl = awp_dpp<0x111>(1.f, awp_row_scan_mul(local));       // product over the channels of the lanes to the left
#pragma unroll
                for (int q = 0; q < 4; ++q) { excl *= om[q]; Q[q] = excl; }
            }
        }
    }
    if (n0 < N) reinterpret_cast<float4*>(out + n * 64)[l16] = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

__global__ __launch_bounds__(256) void k_awp_integrate_bwd_c64(const float* __restrict__ feat, const float* __restrict__ z,
                                                               const float* __restrict__ rays_d, const float* __restrict__ d_out, long N, int S,
                                                               float* __restrict__ d_feat, float* __restrict__ d_z, float* __restrict__ d_rays_d) {
    const int l16 = threadIdx.x & 15;
    const long n0 = blockIdx.x * 16L + (threadIdx.x >> 4);
    const bool live = n0 < N;
    const long n = live ? n0 : N - 1;
    const float* d = rays_d + n * 3;
    const float norm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(d[0], d[0]), __fmul_rn(d[1], d[1])), __fmul_rn(d[2], d[2])));
    const float4* fr = reinterpret_cast<const float4*>(feat + n * (long)S * 64) + l16;
    float4* dfr = reinterpret_cast<float4*>(d_feat + n * (long)S * 64) + l16;
    const float* zz = z + n * (long)S;
    constexpr int PF = 4;                                 // rows per block; the NEXT block's rows are loaded while this one is processed
    float4 cur[PF + 1], nxt[PF];
    const float4 g4 = reinterpret_cast<const float4*>(d_out + n * 64)[l16];
    const float g[4] = {g4.x, g4.y, g4.z, g4.w};
    float Q[4] = {1.f, 1.f, 1.f, 1.f}, ec[4];
#pragma unroll
    for (int u = 0; u <= PF; ++u) cur[u] = fr[(long)min(u, S - 1) * 16];
    {
        const float dist0 = S > 1 ? __fmul_rn(__fsub_rn(zz[1], zz[0]), norm) : 0.f;
        const float f0[4] = {cur[0].x, cur[0].y, cur[0].z, cur[0].w};
#pragma unroll
        for (int q = 0; q < 4; ++q) ec[q] = S > 1 ? awp_exp(-__fmul_rn(f0[q], dist0)) : 1.f;
    }
    float dnorm = 0.f, dz_prev = 0.f;
    for (int s0 = 0; s0 < S; s0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) nxt[u] = fr[(long)min(s0 + PF + 1 + u, S - 1) * 16];
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int s = s0 + u;
            if (s < S) {
                const float fc[4] = {cur[u].x, cur[u].y, cur[u].z, cur[u].w};
                const float fn[4] = {cur[u + 1].x, cur[u + 1].y, cur[u + 1].z, cur[u + 1].w};
                const bool last = s == S - 1;
                const float dz = last ? 0.f : __fsub_rn(zz[s + 1], zz[s]);
                const float dist = __fmul_rn(dz, norm);
                const float dist_n = s + 2 < S ? __fmul_rn(__fsub_rn(zz[s + 2], zz[s + 1]), norm) : 0.f;
                float e[4], en[4], om[4], Qn[4], local = 1.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    e[q] = last ? 1.f : ec[q];
                    om[q] = __fadd_rn(e[q], 1e-10f);                                   // the last row's alpha is 0
                    local *= om[q];
                }
                float excl = awp_dpp<0x111>(1.f, awp_row_scan_mul(local));
#pragma unroll
                for (int q = 0; q < 4; ++q) { excl *= om[q]; Qn[q] = excl; }
                float G[4], lsum = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    en[q] = s + 2 < S ? awp_exp(-__fmul_rn(fn[q], dist_n)) : 1.f;      // e of row s+1 (its alpha is 0 when it is the last row)
                    const float an = s + 2 < S ? __fadd_rn(-en[q], 1.f) : 0.f;
                    G[q] = last ? 0.f : g[q] * an * fn[q] * Qn[q];
                    lsum += G[q];
                }
                float sfx = awp_row_scan_add_right(lsum) - lsum;                        // the lanes to the right
                float ddist = 0.f, df[4];
#pragma unroll
                for (int q = 3; q >= 0; --q) {
                    sfx += G[q];                                                          // channels >= this one
                    const float a = last ? 0.f : __fadd_rn(-e[q], 1.f);
                    const float through = last ? 0.f : sfx * __builtin_amdgcn_rcpf(om[q]);      // (v_rcp_f32, 1 ulp: the IEEE division is ~10 instructions)
                    const float ga = g[q] * Q[q] * fc[q] - through;                       // d out / d a[s,c]
                    df[q] = last ? 0.f : g[q] * Q[q] * a + ga * dist * e[q];
                    ddist += last ? 0.f : ga * fc[q] * e[q];
                }
                if (live) dfr[(long)s * 16] = make_float4(df[0], df[1], df[2], df[3]);
                if (d_z || d_rays_d) {
                    ddist = awp_row_sum(ddist);
                    if (d_z && live && l16 == 0) d_z[n * (long)S + s] = dz_prev - ddist * norm;
                    dz_prev = ddist * norm;
                    dnorm += ddist * dz;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) { Q[q] = Qn[q]; ec[q] = en[q]; }
            }
        }
        cur[0] = cur[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) cur[u + 1] = nxt[u];
    }
    if (d_rays_d && live && l16 < 3) d_rays_d[n * 3 + l16] = norm > 0.f ? dnorm * d[l16] / norm : 0.f;
}

// utils/edi.py:73-95: E_k = -sum_{j=k}^{N-1} bii_j (k<N), 0 (k=N), +sum_{j=N}^{k-1} bii_j (k>N); sharp = (2N+1) blurry / sum exp(E_k)
__global__ void k_edi_deblur(const float* __restrict__ blurry, const float* __restrict__ bii, int steps, long npix,
                             float* __restrict__ sharp) {
    const long px = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (px >= npix) return;
    const int N = (steps - 1) / 2;
    float s = 1.f;       // exp(0) of the frame at f
    float run = 0.f;
    // left part: E_i = -(bii_i + ... + bii_{N-1}); accumulate in the reference's order (i ascending inside each sum)
    for (int i = 0; i < N; ++i) {
        float e = 0.f;
        for (int j = i; j < N; ++j) e += bii[(long)j * npix + px];
        s += expf(-e);
    }
    for (int i = 0; i < N; ++i) {
        run += bii[(long)(N + i) * npix + px];
        s += expf(run);
    }
    sharp[px] = (float)(2 * N + 1) * blurry[px] / s;
}

// utils/edi.py:7-41,44-70: bilinear sub-pixel splat of +-1 events, grey sensor
__global__ void k_edi_splat(const float* __restrict__ x, const float* __restrict__ y, const signed char* __restrict__ p, long n,
                            int w, int h, float c_pos, float c_neg, float* __restrict__ image) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float xv = x[i], yv = y[i];
    const float sc = p[i] > 0 ? c_pos : -c_neg;
#pragma unroll
    for (int xr = 0; xr < 2; ++xr)
#pragma unroll
        for (int yr = 0; yr < 2; ++yr) {
            const float xf = xr ? ceilf(xv) : floorf(xv), yf = yr ? ceilf(yv) : floorf(yv);
            if (!((xf != xv || xr == 0) && (yf != yv || yr == 0) && xf < (float)w && yf < (float)h)) continue;
            const float kx = fmaxf(0.f, 1.f - fabsf(xf - xv)), ky = fmaxf(0.f, 1.f - fabsf(yf - yv));
            atomicAdd(image + (long)yf * w + (long)xf, sc * (kx * ky));
        }
}

}  // namespace evd

using namespace evd;

struct evd_crf {
    CrfParams p;
};

extern "C" {

int evd_weighted_sum(const float* x, const float* ccw, long R, int P, int C, float* out, void* stream) {
    EVD_REQUIRE(R >= 0 && P >= 1 && C >= 1 && out, "evd_weighted_sum: bad arguments");
    if (R == 0) return EVD_OK;
    k_weighted_sum<<<cdiv(R * C, 256), 256, 0, as_stream(stream)>>>(x, ccw, R, P, C, out);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

int evd_crf_create(const evd_crf_desc* d, evd_crf** out) {
    EVD_REQUIRE(d && out, "evd_crf_create: null argument");
    EVD_REQUIRE(d->map_type >= 0 && d->map_type <= 2, "evd_crf_create: map_type %d", d->map_type);
    EVD_REQUIRE(d->extra_features >= 0 && d->extra_features < CRF_MAX_IN, "evd_crf_create: extra_features %d", d->extra_features);
    evd_crf* c = new evd_crf();
    memset(&c->p, 0, sizeof(c->p));
    c->p.map_type = d->map_type;
    c->p.E = d->extra_features;
    c->p.inv_gamma = (float)(1.0 / (double)(d->gamma != 0.f ? d->gamma : 2.2f));   // x ** (1. / gamma), tonemapping.py:68
    if (d->map_type == 2) {
        for (int k = 0; k < 4; ++k)
            if (!d->w[k] || !d->b[k]) { delete c; return fail(EVD_E_INVALID, "evd_crf_create: learn CRF needs 4 weight/bias pairs"); }
        const int nin = 1 + d->extra_features;
        memset(c->p.w0, 0, sizeof(c->p.w0));
        for (int j = 0; j < 16; ++j) memcpy(c->p.w0 + j * CRF_MAX_IN, d->w[0] + j * nin, sizeof(float) * nin);    // rows padded to CRF_MAX_IN
        memcpy(c->p.b0, d->b[0], sizeof(float) * 16);
        memcpy(c->p.w1, d->w[1], sizeof(float) * 256);
        memcpy(c->p.b1, d->b[1], sizeof(float) * 16);
        memcpy(c->p.w2, d->w[2], sizeof(float) * 256);
        memcpy(c->p.b2, d->b[2], sizeof(float) * 16);
        memcpy(c->p.w3, d->w[3], sizeof(float) * 16);
        c->p.b3 = d->b[3][0];
    }
    *out = c;
    return EVD_OK;
}

void evd_crf_destroy(evd_crf* c) { delete c; }

int evd_crf_forward(const evd_crf* crf, const float* x, const float* feat, int feat_per_channel, int skip_learn,
                    int luma, long n, float* out, void* stream) {
    EVD_REQUIRE(crf && n >= 0 && out && luma <= 2, "evd_crf_forward: bad arguments");
    if (n == 0) return EVD_OK;
    k_crf_forward<<<cdiv(n, 256), 256, 0, as_stream(stream)>>>(crf->p, x, feat, feat_per_channel, skip_learn, luma, n, out);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

int evd_blur_loss_reduce(const evd_crf* crf_rgb, int skip_learn, const float* rgb_p, const float* rgb0_p,
                         const float* w1, const float* w2, const float* tgt, const float* tgt0, long R, int P,
                         float* partial, float* out_rgb, float* out_rgb1, float* out_awp, void* stream) {
    EVD_REQUIRE(crf_rgb && rgb_p && w1 && tgt && partial && R >= 0 && P >= 1, "evd_blur_loss_reduce: bad arguments");
    if (R == 0) return EVD_OK;
    k_blur_loss<<<cdiv(3 * R, 64), 64, 0, as_stream(stream)>>>(crf_rgb->p, skip_learn, rgb_p, rgb0_p, w1, w2, tgt, tgt0, R, P, partial,
                                                             out_rgb, out_rgb1, out_awp);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

static int blur_loss_bwd(const evd_crf* crf_rgb, int skip_learn, const float* rgb_p, const float* rgb0_p, const float* w1, const float* w2,
                         const float* tgt, const float* tgt0, long R, int P, const float* g_partial, const float* g_dev, float* d_rgb_p,
                         float* d_rgb0_p, float* d_w1, float* d_w2, void* stream) {
    EVD_REQUIRE(crf_rgb && rgb_p && w1 && tgt && (g_partial || g_dev) && d_rgb_p && R >= 0 && P >= 1, "evd_blur_loss_bwd: bad arguments");
    EVD_REQUIRE(crf_rgb->p.map_type != 2 || skip_learn, "evd_blur_loss_bwd: learnable CRF on the image branch is not built (shipped configs: gamma / none)");
    if (R == 0) return EVD_OK;
    hipStream_t st = as_stream(stream);
    if (d_w1) EVD_HIP(hipMemsetAsync(d_w1, 0, sizeof(float) * (size_t)R * P, st));
    if (d_w2) EVD_HIP(hipMemsetAsync(d_w2, 0, sizeof(float) * (size_t)R * P, st));
    const float zero5[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const float* g = g_partial ? g_partial : zero5;
    k_blur_loss_bwd<<<cdiv(3 * R, 64), 64, 0, st>>>(crf_rgb->p, skip_learn, rgb_p, rgb0_p, w1, w2, tgt, tgt0, R, P, g[0], g[1], g[2], g[3], g[4], g_dev,
                                                    d_rgb_p, rgb0_p ? d_rgb0_p : nullptr, d_w1, d_w2);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

int evd_blur_loss_bwd(const evd_crf* crf_rgb, int skip_learn, const float* rgb_p, const float* rgb0_p, const float* w1, const float* w2,
                      const float* tgt, const float* tgt0, long R, int P, const float* g_partial, float* d_rgb_p, float* d_rgb0_p,
                      float* d_w1, float* d_w2, void* stream) {
    EVD_REQUIRE(g_partial, "evd_blur_loss_bwd: bad arguments");
    return blur_loss_bwd(crf_rgb, skip_learn, rgb_p, rgb0_p, w1, w2, tgt, tgt0, R, P, g_partial, nullptr, d_rgb_p, d_rgb0_p, d_w1, d_w2, stream);
}

int evd_blur_loss_bwd_dev(const evd_crf* crf_rgb, int skip_learn, const float* rgb_p, const float* rgb0_p, const float* w1, const float* w2,
                          const float* tgt, const float* tgt0, long R, int P, const float* g_partial_dev, float* d_rgb_p, float* d_rgb0_p,
                          float* d_w1, float* d_w2, void* stream) {
    EVD_REQUIRE(g_partial_dev, "evd_blur_loss_bwd_dev: bad arguments");
    return blur_loss_bwd(crf_rgb, skip_learn, rgb_p, rgb0_p, w1, w2, tgt, tgt0, R, P, nullptr, g_partial_dev, d_rgb_p, d_rgb0_p, d_w1, d_w2, stream);
}

int evd_event_loss_reduce(const evd_crf* crf_ev, int skip_learn, int add_bii_feat, int tonemap_only,
                          const float* start, const float* end, const float* start0, const float* end0,
                          const float* cum_neg, const float* cum_pos, float thr_neg, float thr_pos,
                          const unsigned char* color_mask, const float* color_weight, long N,
                          float* partial, void* stream) {
    EVD_REQUIRE(crf_ev && start && end && cum_neg && cum_pos && partial && N >= 0, "evd_event_loss_reduce: bad arguments");
    EVD_REQUIRE(add_bii_feat >= 0 && add_bii_feat <= 2, "evd_event_loss_reduce: add_bii_feat %d", add_bii_feat);
    EVD_REQUIRE(add_bii_feat == 0 || crf_ev->p.map_type != 2 || crf_ev->p.E == 2, "evd_event_loss_reduce: bii features need extra_features == 2");
    EVD_REQUIRE(!color_mask || tonemap_only, "evd_event_loss_reduce: a colour mask needs tonemap_only (3-channel luma, utils/events.py:262)");
    EVD_REQUIRE(add_bii_feat != 2 || color_mask, "evd_event_loss_reduce: color-pos-neg features need the colour mask");
    if (N == 0) return EVD_OK;
    const float c0 = color_weight ? color_weight[0] : 1.f, c1 = color_weight ? color_weight[1] : 1.f, c2 = color_weight ? color_weight[2] : 1.f;
    k_event_loss<<<cdiv(N, 16), 256, 0, as_stream(stream)>>>(crf_ev->p, skip_learn, add_bii_feat, tonemap_only, start, end, start0, end0,
                                                              cum_neg, cum_pos, thr_neg, thr_pos, color_mask, c0, c1, c2,
                                                              color_weight != nullptr, N, partial);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

int evd_crf_param_count(void) { return CRF_NPARAM; }

// The handle keeps the (625 + padding) CRF parameters on the HOST (they travel as a kernel argument): get / load exchange them
// with a host array in the gradient layout of evd_event_loss_bwd, so a caller that trains the event-CRF copies 2.8 KB per step.
int evd_crf_get_params(const evd_crf* c, float* host) {
    EVD_REQUIRE(c && host && c->p.map_type == 2, "evd_crf_get_params: needs a learn CRF and a host array");
    float* o = host;
    memcpy(o, c->p.w0, sizeof(c->p.w0)); o += 16 * CRF_MAX_IN;
    memcpy(o, c->p.b0, 64); o += 16;
    memcpy(o, c->p.w1, 1024); o += 256;
    memcpy(o, c->p.b1, 64); o += 16;
    memcpy(o, c->p.w2, 1024); o += 256;
    memcpy(o, c->p.b2, 64); o += 16;
    memcpy(o, c->p.w3, 64); o += 16;
    *o = c->p.b3;
    return EVD_OK;
}

int evd_crf_load_params(evd_crf* c, const float* host) {
    EVD_REQUIRE(c && host && c->p.map_type == 2, "evd_crf_load_params: needs a learn CRF and a host array");
    const float* o = host;
    const int nin = 1 + c->p.E;
    for (int j = 0; j < 16; ++j)
        for (int k = 0; k < CRF_MAX_IN; ++k) c->p.w0[j * CRF_MAX_IN + k] = k < nin ? o[j * CRF_MAX_IN + k] : 0.f;      // padding columns stay zero
    o += 16 * CRF_MAX_IN;
    memcpy(c->p.b0, o, 64); o += 16;
    memcpy(c->p.w1, o, 1024); o += 256;
    memcpy(c->p.b1, o, 64); o += 16;
    memcpy(c->p.w2, o, 1024); o += 256;
    memcpy(c->p.b2, o, 64); o += 16;
    memcpy(c->p.w3, o, 64); o += 16;
    c->p.b3 = *o;
    return EVD_OK;
}

static int event_loss_bwd(const evd_crf* crf_ev, int skip_learn, int add_bii_feat, int tonemap_only,
                          const float* start, const float* end, const float* start0, const float* end0,
                          const float* cum_neg, const float* cum_pos, float thr_neg, float thr_pos,
                          const unsigned char* color_mask, const float* color_weight, long N, float g_fine, float g_coarse, const float* g_dev,
                          float* d_start, float* d_end, float* d_start0, float* d_end0, float* d_params, void* stream) {
    EVD_REQUIRE(crf_ev && start && end && cum_neg && cum_pos && d_start && d_end && N >= 0, "evd_event_loss_bwd: bad arguments");
    EVD_REQUIRE(add_bii_feat >= 0 && add_bii_feat <= 2, "evd_event_loss_bwd: add_bii_feat %d", add_bii_feat);
    EVD_REQUIRE(!color_mask || tonemap_only, "evd_event_loss_bwd: a colour mask needs tonemap_only");
    EVD_REQUIRE(add_bii_feat != 2 || color_mask, "evd_event_loss_bwd: color-pos-neg features need the colour mask");
    hipStream_t st = as_stream(stream);
    if (d_params) EVD_HIP(hipMemsetAsync(d_params, 0, sizeof(float) * CRF_NPARAM, st));
    if (N == 0) return EVD_OK;
    const float c0 = color_weight ? color_weight[0] : 1.f, c1 = color_weight ? color_weight[1] : 1.f, c2 = color_weight ? color_weight[2] : 1.f;
    k_event_loss_bwd<<<cdiv(N, 16), 256, 0, st>>>(crf_ev->p, skip_learn, add_bii_feat, tonemap_only, start, end, start0, end0, cum_neg, cum_pos,
                                                  thr_neg, thr_pos, color_mask, c0, c1, c2, color_weight != nullptr, N, g_fine, g_coarse, g_dev,
                                                  d_start, d_end, (start0 && end0) ? d_start0 : nullptr, (start0 && end0) ? d_end0 : nullptr, d_params);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

int evd_event_loss_bwd(const evd_crf* crf_ev, int skip_learn, int add_bii_feat, int tonemap_only,
                       const float* start, const float* end, const float* start0, const float* end0,
                       const float* cum_neg, const float* cum_pos, float thr_neg, float thr_pos,
                       const unsigned char* color_mask, const float* color_weight, long N, float g_fine, float g_coarse,
                       float* d_start, float* d_end, float* d_start0, float* d_end0, float* d_params, void* stream) {
    return event_loss_bwd(crf_ev, skip_learn, add_bii_feat, tonemap_only, start, end, start0, end0, cum_neg, cum_pos, thr_neg, thr_pos, color_mask,
                          color_weight, N, g_fine, g_coarse, nullptr, d_start, d_end, d_start0, d_end0, d_params, stream);
}

int evd_event_loss_bwd_dev(const evd_crf* crf_ev, int skip_learn, int add_bii_feat, int tonemap_only,
                           const float* start, const float* end, const float* start0, const float* end0,
                           const float* cum_neg, const float* cum_pos, float thr_neg, float thr_pos,
                           const unsigned char* color_mask, const float* color_weight, long N, const float* g_partial_dev,
                           float* d_start, float* d_end, float* d_start0, float* d_end0, float* d_params, void* stream) {
    EVD_REQUIRE(g_partial_dev, "evd_event_loss_bwd_dev: bad arguments");
    return event_loss_bwd(crf_ev, skip_learn, add_bii_feat, tonemap_only, start, end, start0, end0, cum_neg, cum_pos, thr_neg, thr_pos, color_mask,
                          color_weight, N, 0.f, 0.f, g_partial_dev, d_start, d_end, d_start0, d_end0, d_params, stream);
}

int evd_awp_feature_integration(const float* feat, const float* z, const float* rays_d, long N, int S, int C, float* out, void* stream) {
    EVD_REQUIRE(feat && z && rays_d && out && N >= 0 && S >= 1 && C >= 1, "evd_awp_feature_integration: bad arguments");
    EVD_REQUIRE(C <= 256, "evd_awp_feature_integration: %d channels (built: <= 256)", C);
    if (N == 0) return EVD_OK;
    hipStream_t st = as_stream(stream);
    if (C == 64 && !getenv("EVD_AWP_SCAN64")) k_awp_integrate_c64<<<cdiv(N, 16), 256, 0, st>>>(feat, z, rays_d, N, S, out);
    else if (C <= 64) k_awp_integrate<1><<<cdiv(N, 4), 256, 0, st>>>(feat, z, rays_d, N, S, C, out);
    else if (C <= 128) k_awp_integrate<2><<<cdiv(N, 4), 256, 0, st>>>(feat, z, rays_d, N, S, C, out);
    else k_awp_integrate<4><<<cdiv(N, 4), 256, 0, st>>>(feat, z, rays_d, N, S, C, out);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

int evd_awp_feature_integration_bwd(const float* feat, const float* z, const float* rays_d, const float* d_out, long N, int S, int C,
                                    float* d_feat, float* d_z, float* d_rays_d, void* stream) {
    EVD_REQUIRE(feat && z && rays_d && d_out && d_feat && N >= 0 && S >= 1 && C >= 1, "evd_awp_feature_integration_bwd: bad arguments");
    EVD_REQUIRE(C <= 256, "evd_awp_feature_integration_bwd: %d channels (built: <= 256)", C);
    if (N == 0) return EVD_OK;
    hipStream_t st = as_stream(stream);
    if (C == 64 && !getenv("EVD_AWP_SCAN64")) k_awp_integrate_bwd_c64<<<cdiv(N, 16), 256, 0, st>>>(feat, z, rays_d, d_out, N, S, d_feat, d_z, d_rays_d);
    else if (C <= 64) k_awp_integrate_bwd<1><<<cdiv(N, 4), 256, 0, st>>>(feat, z, rays_d, d_out, N, S, C, d_feat, d_z, d_rays_d);
    else if (C <= 128) k_awp_integrate_bwd<2><<<cdiv(N, 4), 256, 0, st>>>(feat, z, rays_d, d_out, N, S, C, d_feat, d_z, d_rays_d);
    else k_awp_integrate_bwd<4><<<cdiv(N, 4), 256, 0, st>>>(feat, z, rays_d, d_out, N, S, C, d_feat, d_z, d_rays_d);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

int evd_edi_deblur(const float* blurry, const float* bii, int steps, long npix, float* sharp, void* stream) {
    EVD_REQUIRE(blurry && bii && sharp && steps >= 3 && (steps & 1) && npix >= 0, "evd_edi_deblur: steps must be odd >= 3");
    if (npix == 0) return EVD_OK;
    k_edi_deblur<<<cdiv(npix, 256), 256, 0, as_stream(stream)>>>(blurry, bii, steps, npix, sharp);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

int evd_edi_bii_image(const float* x, const float* y, const signed char* p, long n, int w, int h,
                      float c_pos, float c_neg, float* image, void* stream) {
    EVD_REQUIRE(image && w > 0 && h > 0 && n >= 0, "evd_edi_bii_image: bad arguments");
    EVD_HIP(hipMemsetAsync(image, 0, sizeof(float) * (size_t)w * h, as_stream(stream)));
    if (n == 0) return EVD_OK;
    k_edi_splat<<<cdiv(n, 256), 256, 0, as_stream(stream)>>>(x, y, p, n, w, h, c_pos, c_neg, image);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

}  // extern "C"
