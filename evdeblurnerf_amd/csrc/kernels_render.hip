// Ray setup, z stratification, alpha compositing scan, inverse-CDF resampling + merge.
// All HBM-bound elementwise / per-ray kernels of the render path (everything except the MLP GEMMs).
#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "evd_common.h"
#include "wave_ops.h"

namespace evd {

constexpr int WAVE = 64;

// ------------------------------------------------------------------------------------------------
// wave-level helpers (64-wide wavefronts)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
    return v;
}
// inclusive product scan over the 64 lanes
__device__ __forceinline__ float wave_scan_mul(float v, int lane) {
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
        float o = __shfl_up(v, off, WAVE);
        if (lane >= off) v *= o;
    }
    return v;
}
__device__ __forceinline__ double wave_scan_add_d(double v, int lane) {
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
        double o = __shfl_up(v, off, WAVE);
        if (lane >= off) v += o;
    }
    return v;
}

// ------------------------------------------------------------------------------------------------
// reference utils/rays.py:8-22
struct Pose34 { float m[12]; };
__global__ void k_get_rays(int H, int W, float k00, float k02, float k11, float k12, float halfpix, const Pose34 pose,
                           float* __restrict__ ro, float* __restrict__ rd) {
    const float* c2w = pose.m;
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (idx >= (long)H * W) return;
    const int j = idx / W, i = idx % W;
    const float d0 = ((float)i + (halfpix - k02)) / k00, d1 = -((float)j + (halfpix - k12)) / k11, d2 = -1.f;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        rd[idx * 3 + r] = __fadd_rn(__fadd_rn(__fmul_rn(d0, c2w[r * 4]), __fmul_rn(d1, c2w[r * 4 + 1])), __fmul_rn(d2, c2w[r * 4 + 2]));
        ro[idx * 3 + r] = c2w[r * 4 + 3];
    }
}

// reference utils/rays.py:25-36
__global__ void k_get_rays_pix(const float* __restrict__ coords, float k00, float k02, float k11, float k12, float halfpix,
                               const float* __restrict__ c2ws, long n, float* __restrict__ ro, float* __restrict__ rd) {
    const long p = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (p >= n) return;
    const float* c2w = c2ws + p * 12;
    const float d0 = (coords[p * 2] + (halfpix - k02)) / k00, d1 = -(coords[p * 2 + 1] + (halfpix - k12)) / k11, d2 = -1.f;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        rd[p * 3 + r] = __fadd_rn(__fadd_rn(__fmul_rn(d0, c2w[r * 4]), __fmul_rn(d1, c2w[r * 4 + 1])), __fmul_rn(d2, c2w[r * 4 + 2]));
        ro[p * 3 + r] = c2w[r * 4 + 3];
    }
}

// reference utils/rays.py:104-145 (unfused mul/add like the reference's elementwise tensor ops)
__device__ __forceinline__ void ndc_one(float cw, float ch, float near, float two_near, const float o[3], const float d[3],
                                        float oo[3], float od[3]) {
    const float t = -__fadd_rn(near, o[2]) / d[2];
    const float ox = __fadd_rn(o[0], __fmul_rn(t, d[0])), oy = __fadd_rn(o[1], __fmul_rn(t, d[1])),
                oz = __fadd_rn(o[2], __fmul_rn(t, d[2]));
    const float ox_oz = ox / oz, oy_oz = oy / oz;
    const float o2 = __fadd_rn(1.f, two_near / oz);
    oo[0] = __fmul_rn(cw, ox_oz);
    oo[1] = __fmul_rn(ch, oy_oz);
    oo[2] = o2;
    od[0] = __fmul_rn(cw, __fsub_rn(d[0] / d[2], ox_oz));
    od[1] = __fmul_rn(ch, __fsub_rn(d[1] / d[2], oy_oz));
    od[2] = __fsub_rn(1.f, o2);
}

__global__ void k_ndc(float cw, float ch, float near, float two_near, const float* __restrict__ o, const float* __restrict__ d,
                      long n, float* __restrict__ oo, float* __restrict__ od) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a[3] = {o[i * 3], o[i * 3 + 1], o[i * 3 + 2]}, b[3] = {d[i * 3], d[i * 3 + 1], d[i * 3 + 2]}, x[3], y[3];
    ndc_one(cw, ch, near, two_near, a, b, x, y);
#pragma unroll
    for (int c = 0; c < 3; ++c) { oo[i * 3 + c] = x[c]; od[i * 3 + c] = y[c]; }
}

// RigidBlurringModel.rbk_warp (networks/dpnerf/blurmodel.py:51-82; SE3Field.get_transform / warp, RigidBody.exp_se3 /
// exp_so3 of utils/rigid_warping.py:18-49,72-107): one thread per (ray, sub-exposure slot).  The reference runs ~40 small
// tensor ops per motion in a Python loop over the 9 motions.
__global__ void k_rbk_warp(const float* __restrict__ rays, const float* __restrict__ r, const float* __restrict__ v, long R, int M,
                           int use_origin, float* __restrict__ new_rays, float* __restrict__ transforms) {
    const int P = M + (use_origin ? 1 : 0);
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (idx >= R * P) return;
    const long n = idx / P;
    const int slot = idx % P, i = slot - (use_origin ? 1 : 0);
    float o[3], d[3], e[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { o[c] = rays[n * 6 + c * 2]; d[c] = rays[n * 6 + c * 2 + 1]; e[c] = __fadd_rn(o[c], d[c]); }
    float T[16] = {1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 1.f};
    float wo[3] = {o[0], o[1], o[2]}, wd[3] = {d[0], d[1], d[2]};
    if (i >= 0) {
        const float rot[3] = {r[(n * 3 + 0) * M + i], r[(n * 3 + 1) * M + i], r[(n * 3 + 2) * M + i]};
        const float tr[3] = {v[(n * 3 + 0) * M + i], v[(n * 3 + 1) * M + i], v[(n * 3 + 2) * M + i]};
        const float theta = __fadd_rn(sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(rot[0], rot[0]), __fmul_rn(rot[1], rot[1])), __fmul_rn(rot[2], rot[2]))), 1.0e-10f);
        const float w[3] = {rot[0] / theta, rot[1] / theta, rot[2] / theta}, vv[3] = {tr[0] / theta, tr[1] / theta, tr[2] / theta};
        const float Wm[9] = {0.f, -w[2], w[1], w[2], 0.f, -w[0], -w[1], w[0], 0.f};
        float W2[9];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 3; ++k) acc = __fadd_rn(acc, __fmul_rn(Wm[a * 3 + k], Wm[k * 3 + b]));
                W2[a * 3 + b] = acc;
            }
        const float st = sinf(theta), omc = __fsub_rn(1.0f, cosf(theta)), tms = __fsub_rn(theta, st);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float p = 0.f;
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const float eye = a == b ? 1.f : 0.f;
                T[a * 4 + b] = __fadd_rn(__fadd_rn(eye, __fmul_rn(st, Wm[a * 3 + b])), __fmul_rn(omc, W2[a * 3 + b]));
                const float g = __fadd_rn(__fadd_rn(__fmul_rn(theta, eye), __fmul_rn(omc, Wm[a * 3 + b])), __fmul_rn(tms, W2[a * 3 + b]));
                p = __fadd_rn(p, __fmul_rn(g, vv[b]));
            }
            T[a * 4 + 3] = p;
        }
        float we[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            wo[a] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[a * 4], o[0]), __fmul_rn(T[a * 4 + 1], o[1])), __fmul_rn(T[a * 4 + 2], o[2])), T[a * 4 + 3]);
            we[a] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[a * 4], e[0]), __fmul_rn(T[a * 4 + 1], e[1])), __fmul_rn(T[a * 4 + 2], e[2])), T[a * 4 + 3]);
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) wd[a] = __fsub_rn(we[a], wo[a]);
    }
    float* out = new_rays + idx * 6;
#pragma unroll
    for (int c = 0; c < 3; ++c) { out[c * 2] = wo[c]; out[c * 2 + 1] = wd[c]; }
    if (transforms) {
#pragma unroll
        for (int k = 0; k < 16; ++k) transforms[idx * 16 + k] = T[k];
    }
}

// reference networks/renderer.py:423-446: rays [R,3,2] -> ray_batch [R,ncol]
__global__ void k_ray_batch(const float* __restrict__ rays, long R, int ndc, int use_viewdirs, float cw, float ch,
                            float near, float far, float* __restrict__ rb) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= R) return;
    const int nc = use_viewdirs ? 11 : 8;
    float o[3], d[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { o[c] = rays[i * 6 + c * 2]; d[c] = rays[i * 6 + c * 2 + 1]; }
    float* out = rb + i * nc;
    if (use_viewdirs) {
        const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(d[0], d[0]), __fmul_rn(d[1], d[1])), __fmul_rn(d[2], d[2])));
#pragma unroll
        for (int c = 0; c < 3; ++c) out[8 + c] = d[c] / nrm;
    }
    if (ndc) {
        float x[3], y[3];
        ndc_one(cw, ch, 1.f, 2.f, o, d, x, y);   // get_ndc_rays(H, W, K[0][0], 1., ...) renderer.py:437
#pragma unroll
        for (int c = 0; c < 3; ++c) { o[c] = x[c]; d[c] = y[c]; }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) { out[c] = o[c]; out[3 + c] = d[c]; }
    out[6] = near;
    out[7] = far;
}

// Backward of k_ray_batch (training: the loss reaches the blur kernel's camera motion through the packed rays, renderer.py:303-308):
// d ray_batch [R,11] -> d rays [R,3,2].  One thread per ray; the forward's intermediate values are recomputed.  near / far columns carry
// no gradient.  With o' = o + t d, t = -(1 + o_z) / d_z, a = o'_z:   o_out = (cw o'_x / a, ch o'_y / a, 1 + 2 / a),
// d_out = (cw (d_x / d_z - o'_x / a), ch (d_y / d_z - o'_y / a), -2 / a),   viewdirs = d / |d|   (utils/rays.py:104-145, renderer.py:431).
__global__ void k_ray_batch_bwd(const float* __restrict__ rays, const float* __restrict__ g, long R, int ndc, float cw, float ch, float* __restrict__ d_rays) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= R) return;
    float o[3], d[3], go[3], gd[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { o[c] = rays[i * 6 + c * 2]; d[c] = rays[i * 6 + c * 2 + 1]; }
    const float* gr = g + i * 11;
    const float goo[3] = {gr[0], gr[1], gr[2]}, gdo[3] = {gr[3], gr[4], gr[5]}, gv[3] = {gr[8], gr[9], gr[10]};
    if (ndc) {
        const float t = -(1.f + o[2]) / d[2];
        const float px = o[0] + t * d[0], py = o[1] + t * d[1], a = o[2] + t * d[2];
        const float ia = 1.f / a, idz = 1.f / d[2];
        const float g_ox = cw * (goo[0] - gdo[0]), g_oy = ch * (goo[1] - gdo[1]);            // d loss / d (o'_x / a), d (o'_y / a)
        const float gp[3] = {g_ox * ia, g_oy * ia, (2.f * (gdo[2] - goo[2]) - g_ox * px - g_oy * py) * ia * ia};   // d loss / d o'
        const float g_t = gp[0] * d[0] + gp[1] * d[1] + gp[2] * d[2];
        go[0] = gp[0]; go[1] = gp[1]; go[2] = gp[2] - g_t * idz;
        gd[0] = t * gp[0] + cw * gdo[0] * idz;
        gd[1] = t * gp[1] + ch * gdo[1] * idz;
        gd[2] = t * gp[2] - (cw * gdo[0] * d[0] + ch * gdo[1] * d[1]) * idz * idz + g_t * (1.f + o[2]) * idz * idz;
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) { go[c] = goo[c]; gd[c] = gdo[c]; }
    }
    const float n2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2], inr = 1.f / sqrtf(n2);
    const float dot = (gv[0] * d[0] + gv[1] * d[1] + gv[2] * d[2]) / n2;                    // (viewdirs . g) / |d|
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        gd[c] += (gv[c] - d[c] * dot) * inr;
        d_rays[i * 6 + c * 2] = go[c];
        d_rays[i * 6 + c * 2 + 1] = gd[c];
    }
}

// pts = o + d z under autograd: d pts [R,S,3] -> rows of d ray_batch (columns 0..2 += sum_s d pts, 3..5 += sum_s z d pts; the others are
// left alone): one wavefront per ray, DPP-free shuffle reduction (S <= a few hundred)
__global__ __launch_bounds__(256) void k_points_bwd(const float* __restrict__ z, const float* __restrict__ d_pts, long R, int S, int accumulate,
                                                    float* __restrict__ d_rb) {
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= R) return;
    float a[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int s = lane; s < S; s += 64) {
        const float zv = z[r * S + s];
        const float* p = d_pts + (r * S + s) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) { a[c] += p[c]; a[3 + c] += zv * p[c]; }
    }
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) a[q] += __shfl_xor(a[q], off);
    if (lane < 11) {                 // overwrite mode: the whole row (zeros in the near / far / viewdirs columns), so the caller needs no fill
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 6; ++q) v = lane == q ? a[q] : v;
        float* out = d_rb + r * 11 + lane;
        if (accumulate) { if (lane < 6) *out += v; }
        else *out = v;
    }
}

// reference networks/renderer.py:163-178
__global__ void k_sample_z(const float* __restrict__ rb, int nc, long R, int S, int lindisp, int perturb,
                           const float* __restrict__ t_rand, float* __restrict__ z, float* __restrict__ pts) {
    // pts (or null): the sample positions o + d z as k_points writes them (renderer.py:180), one launch less in a c2f render
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (idx >= R * S) return;
    const long r = idx / S;
    const int i = idx % S;
    const float near = rb[r * nc + 6], far = rb[r * nc + 7];
    auto zat = [&](int k) {
        const float t = linspace_at(0.f, 1.f, S, k);
        return lindisp ? 1.f / __fadd_rn(__fmul_rn(1.f / near, __fsub_rn(1.f, t)), __fmul_rn(1.f / far, t))
                       : __fadd_rn(__fmul_rn(near, __fsub_rn(1.f, t)), __fmul_rn(far, t));
    };
    float zi = zat(i);
    if (perturb) {
        const float upper = (i < S - 1) ? __fmul_rn(.5f, __fadd_rn(zat(i + 1), zi)) : zi;
        const float lower = (i > 0) ? __fmul_rn(.5f, __fadd_rn(zi, zat(i - 1))) : zi;
        zi = __fadd_rn(lower, __fmul_rn(__fsub_rn(upper, lower), t_rand[idx]));
    }
    z[idx] = zi;
    if (pts) {
#pragma unroll
        for (int c = 0; c < 3; ++c) pts[idx * 3 + c] = __fadd_rn(rb[r * nc + c], __fmul_rn(rb[r * nc + 3 + c], zi));
    }
}

// k_ray_batch + k_sample_z in one launch (evd_nerf_render: near / far are the configuration's scalars, so z does not depend on the packed
// row): thread = (ray, sample); the sample-0 thread also packs the ray's row.  Same arithmetic as the two kernels.
__global__ void k_ray_batch_z(const float* __restrict__ rays, long R, int ndc, float cw, float ch, float near, float far, int S, int lindisp, int perturb,
                              const float* __restrict__ t_rand, float* __restrict__ rb, float* __restrict__ z) {
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (idx >= R * S) return;
    const long r = idx / S;
    const int i = idx % S;
    auto zat = [&](int k) {
        const float t = linspace_at(0.f, 1.f, S, k);
        return lindisp ? 1.f / __fadd_rn(__fmul_rn(1.f / near, __fsub_rn(1.f, t)), __fmul_rn(1.f / far, t))
                       : __fadd_rn(__fmul_rn(near, __fsub_rn(1.f, t)), __fmul_rn(far, t));
    };
    float zi = zat(i);
    if (perturb) {
        const float upper = (i < S - 1) ? __fmul_rn(.5f, __fadd_rn(zat(i + 1), zi)) : zi;
        const float lower = (i > 0) ? __fmul_rn(.5f, __fadd_rn(zi, zat(i - 1))) : zi;
        zi = __fadd_rn(lower, __fmul_rn(__fsub_rn(upper, lower), t_rand[idx]));
    }
    z[idx] = zi;
    if (i != 0) return;
    float o[3], d[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { o[c] = rays[r * 6 + c * 2]; d[c] = rays[r * 6 + c * 2 + 1]; }
    float* out = rb + r * 11;
    const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(d[0], d[0]), __fmul_rn(d[1], d[1])), __fmul_rn(d[2], d[2])));
#pragma unroll
    for (int c = 0; c < 3; ++c) out[8 + c] = d[c] / nrm;
    if (ndc) {
        float x[3], y[3];
        ndc_one(cw, ch, 1.f, 2.f, o, d, x, y);   // get_ndc_rays(H, W, K[0][0], 1., ...) renderer.py:437
#pragma unroll
        for (int c = 0; c < 3; ++c) { o[c] = x[c]; d[c] = y[c]; }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) { out[c] = o[c]; out[3 + c] = d[c]; }
    out[6] = near;
    out[7] = far;
}

// reference networks/embedding.py:88-98
__global__ void k_embed(const float* __restrict__ x, long n, int dim, int L, float* __restrict__ out) {
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (idx >= n * dim) return;
    const long i = idx / dim;
    const int c = idx % dim;
    const int od = dim * (1 + 2 * L);
    const float v = x[idx];
    float* o = out + i * od;
    o[c] = v;
    float f = 1.f;
    for (int k = 0; k < L; ++k) {
        const float a = v * f;
        o[dim + 2 * k * dim + c] = sinf(a);
        o[dim + (2 * k + 1) * dim + c] = cosf(a);
        f *= 2.f;
    }
}

// ------------------------------------------------------------------------------------------------
// raw2outputs: one wavefront per ray, 64 samples per pass, transmittance carried across passes.
// HBM-bound: reads raw (4C B) + z (4 B), writes weights (4 B) per sample.  reference networks/nerf.py:74-129,
// networks/pdrf/voxnerf.py:153-201: density on the first S-1 samples, last alpha forced to 1,
// T = exclusive cumprod(1 - alpha) done as a wavefront product scan.
template <int NRGB>
__global__ __launch_bounds__(256) void k_composite(const float* __restrict__ raw, const float* __restrict__ z,
                                                   const float* __restrict__ rays_d, int rd_stride, long R, int S, int C,
                                                   int sigma_ch, int rgb_ch0, int n_rgb, int rgb_act, int sigma_act,
                                                   int white_bkgd, float rmnear, const float* __restrict__ noise,
                                                   float* __restrict__ out_map, float* __restrict__ density,
                                                   float* __restrict__ acc, float* __restrict__ weights,
                                                   float* __restrict__ depth) {
    const int lane = threadIdx.x & 63;
    const long r = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= R) return;
    const float* d = rays_d + r * rd_stride;
    const float norm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(d[0], d[0]), __fmul_rn(d[1], d[1])), __fmul_rn(d[2], d[2])));
    const float* rw = raw + r * (long)S * C;
    const float* zz = z + r * (long)S;
    float carry = 1.f, a_sum = 0.f, d_sum = 0.f;
    float csum[NRGB > 0 ? NRGB : 1];
#pragma unroll
    for (int c = 0; c < NRGB; ++c) csum[c] = 0.f;
    for (int base = 0; base < S; base += 64) {
        const int i = base + lane;
        const bool valid = i < S;
        float alpha = 0.f, zi = 0.f;
        float rgbv[NRGB > 0 ? NRGB : 1];
        if (valid) {
            zi = zz[i];
            float sraw;
            if (NRGB == 3 && C == 4) {   // fast path: one 16-byte load per sample
                const float4 v = reinterpret_cast<const float4*>(rw)[i];
                const float vv[4] = {v.x, v.y, v.z, v.w};
                sraw = vv[sigma_ch];
#pragma unroll
                for (int c = 0; c < NRGB; ++c) rgbv[c] = vv[rgb_ch0 + c];
            } else {
                sraw = rw[(long)i * C + sigma_ch];
#pragma unroll
                for (int c = 0; c < NRGB; ++c) rgbv[c] = rw[(long)i * C + rgb_ch0 + c];
            }
            if (i < S - 1) {
                const float dist = __fmul_rn(__fsub_rn(zz[i + 1], zi), norm);
                if (noise) sraw = __fadd_rn(sraw, noise[r * (long)(S - 1) + i]);
                float dens = act(sigma_act, sraw);
                if (rmnear > 0.f) dens = (zz[i + 1] > rmnear ? 1.f : 0.f) * dens;
                if (density) density[r * (long)(S - 1) + i] = dens;
                alpha = __fadd_rn(-expf(-__fmul_rn(dens, dist)), 1.f);
            } else {
                alpha = 1.f;
            }
        }
        const float om = valid ? __fadd_rn(-alpha, 1.f) : 1.f;
        const float incl = wave_scan_mul(om, lane);
        float excl = __shfl_up(incl, 1, WAVE);
        if (lane == 0) excl = 1.f;
        const float T = carry * excl;
        const float w = valid ? alpha * T : 0.f;
        carry *= __shfl(incl, 63, WAVE);
        if (valid && weights) weights[r * (long)S + i] = w;
        a_sum += w;
        d_sum += w * zi;
#pragma unroll
        for (int c = 0; c < NRGB; ++c) csum[c] += valid ? w * act(rgb_act, rgbv[c]) : 0.f;
    }
    a_sum = wave_sum(a_sum);
    d_sum = wave_sum(d_sum);
#pragma unroll
    for (int c = 0; c < NRGB; ++c) csum[c] = wave_sum(csum[c]);
    if (lane == 0) {
        if (acc) acc[r] = a_sum;
        if (depth) depth[r] = d_sum;
        if (out_map) {
#pragma unroll
            for (int c = 0; c < NRGB; ++c) out_map[r * n_rgb + c] = white_bkgd ? csum[c] + (1.f - a_sum) : csum[c];
        }
    }
}

// raw2outputs, bandwidth form (C == 4, three colour channels, S <= 64 * SPL): one wavefront per ray, every lane owns
// SPL CONSECUTIVE samples, so all of a ray's loads (SPL float4 of raw + SPL floats of z per lane, 2.5 KiB per
// wavefront at S = 128) are issued before the first dependent instruction; the transmittance is a lane-local
// product followed by one DPP scan of the 64 lane totals.  Same arithmetic per sample as k_composite.
template <int SPL, int RPW>
__global__ __launch_bounds__(256) void k_composite_rows(const float* __restrict__ raw, const float* __restrict__ z,
                                                        const float* __restrict__ rays_d, int rd_stride, long R, int S,
                                                        int sigma_ch, int rgb_ch0, int rgb_act, int sigma_act, int white_bkgd,
                                                        float rmnear, const float* __restrict__ noise,
                                                        float* __restrict__ out_map, float* __restrict__ density,
                                                        float* __restrict__ acc, float* __restrict__ weights, float* __restrict__ depth) {
    const int lane = threadIdx.x & 63;
    const long r0 = (blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6)) * RPW;   // RPW consecutive rays per wavefront
    if (r0 >= R) return;
    const int i0 = lane * SPL;
    float4 v[RPW][SPL];
    float zi[RPW][SPL + 1];
    float dd[RPW][3];
#pragma unroll
    for (int q = 0; q < RPW; ++q) {             // every load of the wavefront's rays first
        const long r = min(r0 + q, R - 1);
        const float4* rw = reinterpret_cast<const float4*>(raw + r * (long)S * 4);
        const float* zz = z + r * (long)S;
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            const int i = min(i0 + j, S - 1);
            v[q][j] = rw[i];
            zi[q][j] = zz[i];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) dd[q][c] = rays_d[r * rd_stride + c];
    }
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const long r = r0 + q;
        if (r >= R) break;
        const float* d = dd[q];
        const float norm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(d[0], d[0]), __fmul_rn(d[1], d[1])), __fmul_rn(d[2], d[2])));
        zi[q][SPL] = dpp_f32<0x130>(0.f, zi[q][0]);          // wave_shl:1 -- the next lane's first z
        float alpha[SPL], om[SPL];
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            const int i = i0 + j;
            const float vv[4] = {v[q][j].x, v[q][j].y, v[q][j].z, v[q][j].w};
            float sraw = vv[sigma_ch];
            if (i < S - 1) {
                const float dist = __fmul_rn(__fsub_rn(zi[q][j + 1], zi[q][j]), norm);
                if (noise) sraw = __fadd_rn(sraw, noise[r * (long)(S - 1) + i]);
                float dens = act_fast(sigma_act, sraw);
                if (rmnear > 0.f) dens = (zi[q][j + 1] > rmnear ? 1.f : 0.f) * dens;
                if (density) density[r * (long)(S - 1) + i] = dens;
                alpha[j] = __fadd_rn(-expf(-__fmul_rn(dens, dist)), 1.f);
            } else {
                alpha[j] = i == S - 1 ? 1.f : 0.f;     // last sample: alpha forced to 1 (nerf.py:113-114); beyond S: nothing
            }
            om[j] = i < S ? __fadd_rn(-alpha[j], 1.f) : 1.f;
        }
        float local = om[0];
#pragma unroll
        for (int j = 1; j < SPL; ++j) local *= om[j];
        const float incl = wave_scan_mul_dpp(local);
        float T = dpp_f32<0x138>(1.f, incl);           // wave_shr:1 -> exclusive product; lane 0 keeps 1
        float a_sum = 0.f, d_sum = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
        float w[SPL];
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            const float vv[4] = {v[q][j].x, v[q][j].y, v[q][j].z, v[q][j].w};
            w[j] = (i0 + j < S) ? alpha[j] * T : 0.f;
            T *= om[j];
            a_sum += w[j];
            d_sum += w[j] * zi[q][j];
            c0 += w[j] * act_fast(rgb_act, vv[rgb_ch0]);
            c1 += w[j] * act_fast(rgb_act, vv[rgb_ch0 + 1]);
            c2 += w[j] * act_fast(rgb_act, vv[rgb_ch0 + 2]);
        }
        if (weights) {
            float* wo = weights + r * (long)S + i0;
            if (SPL == 2 && (S & 1) == 0) {
                if (i0 < S) *reinterpret_cast<float2*>(wo) = make_float2(w[0], w[1]);
            } else if (SPL == 4 && (S & 3) == 0) {
                if (i0 < S) *reinterpret_cast<float4*>(wo) = make_float4(w[0], w[1], w[2], w[3]);
            } else {
#pragma unroll
                for (int j = 0; j < SPL; ++j) if (i0 + j < S) wo[j] = w[j];
            }
        }
        a_sum = wave_sum_dpp(a_sum);
        d_sum = wave_sum_dpp(d_sum);
        c0 = wave_sum_dpp(c0);
        c1 = wave_sum_dpp(c1);
        c2 = wave_sum_dpp(c2);
        if (lane == 0) {
            if (acc) acc[r] = a_sum;
            if (depth) depth[r] = d_sum;
            if (out_map) {
                const float bg = white_bkgd ? 1.f - a_sum : 0.f;
                out_map[r * 3] = white_bkgd ? c0 + bg : c0;
                out_map[r * 3 + 1] = white_bkgd ? c1 + bg : c1;
                out_map[r * 3 + 2] = white_bkgd ? c2 + bg : c2;
            }
        }
    }
}

typedef float cs_f4 __attribute__((ext_vector_type(4)));     // native vectors: the non-temporal builtins do not take HIP's float4 / float2

// raw2outputs, interleaved form (round 5): lane l owns samples l, 64 + l, ... (NCH chunks of 64), so that every load / store instruction
// of the wavefront covers ONE contiguous run (k_composite_rows' lane owns consecutive samples: its 16-byte loads are 32 B apart and every
// 128-byte line is requested by two instructions).  The transmittance is one DPP product scan per chunk with the running product carried
// across chunks; channel layout and activations are template constants (k_composite_rows selects them at run time: a select chain per
// dynamic register index and a switch per activation).  All loads of the wavefront's RPW rays are issued before the first dependent
// instruction; the streamed raw / z / weights rows move with non-temporal accesses (NT).  A persistent form with the next group's loads
// in flight under the current group's arithmetic (k_composite_stream) measured 0.55 of 8 TB/s against 0.64 (rows) and 0.72 (this): removed.
// sum over the 64 lanes, valid in LANE 63 only (row sums, then the two row broadcasts of the scan): 6 DPP adds, no readlane
__device__ __forceinline__ float wave_sum_to63(float v) {
    v += dpp_f32<0xb1>(0.f, v);
    v += dpp_f32<0x4e>(0.f, v);
    v += dpp_f32<0x141>(0.f, v);
    v += dpp_f32<0x140>(0.f, v);
    v += dpp_f32<0x142, 0xa>(0.f, v);
    v += dpp_f32<0x143, 0xc>(0.f, v);
    return v;
}

template <int NCH, int RPW, int SIGMA_CH, int RGB_ACT, int SIGMA_ACT, bool NT>
__global__ __launch_bounds__(256) void k_composite_il(const float* __restrict__ raw, const float* __restrict__ z,
                                                      const float* __restrict__ rays_d, int rd_stride, long R, int S, int white_bkgd,
                                                      float* __restrict__ out_map, float* __restrict__ acc, float* __restrict__ weights,
                                                      float* __restrict__ depth, const float* __restrict__ noise, float* __restrict__ density) {
    constexpr int RGB0 = SIGMA_CH == 0 ? 1 : 0;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);              // scalar: the ray index and every row base stay in SGPRs
    const long r0 = (blockIdx.x * (long)(blockDim.x >> 6) + wave) * RPW;
    if (r0 >= R) return;
    cs_f4 v[RPW][NCH];
    float zi[RPW][NCH];
    float dd[RPW][3];
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const long r = min(r0 + q, R - 1);
        const cs_f4* rw = reinterpret_cast<const cs_f4*>(raw + r * (long)S * 4);
        const float* zz = z + r * (long)S;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int i = c + 1 < NCH ? c * 64 + lane : min(c * 64 + lane, S - 1);     // only the last chunk can run past the row
            v[q][c] = NT ? __builtin_nontemporal_load(rw + i) : rw[i];
            zi[q][c] = NT ? __builtin_nontemporal_load(zz + i) : zz[i];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) dd[q][c] = rays_d[r * rd_stride + c];
    }
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const long r = r0 + q;
        if (r >= R) break;
        const float* d = dd[q];
        const float norm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(d[0], d[0]), __fmul_rn(d[1], d[1])), __fmul_rn(d[2], d[2])));
        float carry = 1.f, a_sum = 0.f, d_sum = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int i = c * 64 + lane;
            // the next sample's z: the next lane's, and for lane 63 the next chunk's lane 0
            const float z_up = c + 1 < NCH ? __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, zi[q][c + 1 < NCH ? c + 1 : 0]))) : 0.f;
            const float zn = dpp_f32<0x130>(z_up, zi[q][c]);              // wave_shl:1, lane 63 keeps `old`
            const cs_f4 v4 = v[q][c];
            float sraw = SIGMA_CH == 0 ? v4.x : v4.w;
            const float dist = __fmul_rn(__fsub_rn(zn, zi[q][c]), norm);
            // the training node's optional extras (nerf.py:106-111 raw noise; the activated density it returns): S - 1 columns per ray
            if (noise && i < S - 1) sraw = __fadd_rn(sraw, noise[r * (long)(S - 1) + i]);
            const float dens = act_fast(SIGMA_ACT, sraw);
            if (density && i < S - 1) density[r * (long)(S - 1) + i] = dens;
            // 1 - exp(-sigma dist) with the hardware exp2 (absolute error of alpha <= 1e-7, the size of its own float32 rounding)
            float alpha = __fadd_rn(-__expf(-__fmul_rn(dens, dist)), 1.f);
            float om = __fadd_rn(-alpha, 1.f);
            if (c + 1 == NCH) {                                           // the row's last sample (alpha forced to 1, nerf.py:113-114) and the lanes behind it
                alpha = i < S - 1 ? alpha : (i == S - 1 ? 1.f : 0.f);
                om = i < S - 1 ? om : (i == S - 1 ? 0.f : 1.f);
            }
            const float incl = wave_scan_mul_dpp(om);
            const float T = carry * dpp_f32<0x138>(1.f, incl);            // wave_shr:1 -> exclusive product; lane 0 keeps 1
            if (c + 1 < NCH) carry *= __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, incl), 63));
            const float w = alpha * T;                                    // (alpha = 0 behind the row's end)
            if (weights && (c + 1 < NCH || i < S)) {
                float* wo = weights + r * (long)S + i;
                if (NT) __builtin_nontemporal_store(w, wo); else *wo = w;
            }
            a_sum += w;
            d_sum += w * zi[q][c];
            c0 += w * act_fast(RGB_ACT, RGB0 ? v4.y : v4.x);
            c1 += w * act_fast(RGB_ACT, RGB0 ? v4.z : v4.y);
            c2 += w * act_fast(RGB_ACT, RGB0 ? v4.w : v4.z);
        }
        a_sum = wave_sum_to63(a_sum);
        d_sum = wave_sum_to63(d_sum);
        c0 = wave_sum_to63(c0);
        c1 = wave_sum_to63(c1);
        c2 = wave_sum_to63(c2);
        if (lane == 63) {
            if (acc) acc[r] = a_sum;
            if (depth) depth[r] = d_sum;
            if (out_map) {
                const float bg = white_bkgd ? 1.f - a_sum : 0.f;
                out_map[r * 3] = white_bkgd ? c0 + bg : c0;
                out_map[r * 3 + 1] = white_bkgd ? c1 + bg : c1;
                out_map[r * 3 + 2] = white_bkgd ? c2 + bg : c2;
            }
        }
    }
}

// derivative of evd::act / act_fast at x (y = the activation's value where that is cheaper)
__device__ __forceinline__ float act_grad(int code, float x) {
    switch (code) {
    case EVD_ACT_RELU: return x > 0.f ? 1.f : 0.f;
    case EVD_ACT_SIGMOID: { const float y = 1.f / (1.f + expf(-x)); return y * (1.f - y); }
    case EVD_ACT_EXP: return expf(x);
    case EVD_ACT_SIGMOID1: { const float y = 1.f / (expf(-x) + 1.f); return 1.002f * y * (1.f - y); }
    case EVD_ACT_SOFTPLUS: { const float t = x - 1.f; return t > 20.f ? 1.f : 1.f / (1.f + expf(-t)); }
    case EVD_ACT_TANH: { const float y = tanhf(x); return 1.f - y * y; }
    default: return 1.f;
    }
}

__device__ __forceinline__ float wave_scan_add_dpp(float v) {
    v += dpp_f32<0x111>(0.f, v);
    v += dpp_f32<0x112>(0.f, v);
    v += dpp_f32<0x114>(0.f, v);
    v += dpp_f32<0x118>(0.f, v);
    v += dpp_f32<0x142, 0xa>(0.f, v);
    v += dpp_f32<0x143, 0xc>(0.f, v);
    return v;
}

// Backward of raw2outputs ("next" row f-1, first slice): d raw from the gradients of (out_map, depth, acc, weights).
// Same decomposition as k_composite_rows (a wavefront per ray, SPL consecutive samples per lane); the forward quantities
// are recomputed.  With G_i = dL/dw_i = g_map . rgb_i + g_depth z_i + g_acc + g_w_i (- sum g_map for a white background):
//   dL/d rgb_i = g_map w_i,      dL/d sigma_i = dist_i (G_i T_{i+1} - sum_{j>i} G_j w_j)      (no division: T_i (1 - alpha_i) = T_{i+1})
// the suffix sum is total - inclusive prefix (one DPP add scan); the last sample's alpha is the constant 1.
template <int SPL>
__global__ __launch_bounds__(256) void k_composite_rows_bwd(const float* __restrict__ raw, const float* __restrict__ z,
                                                            const float* __restrict__ rays_d, int rd_stride, long R, int S,
                                                            int sigma_ch, int rgb_ch0, int rgb_act, int sigma_act, int white_bkgd,
                                                            float rmnear, const float* __restrict__ noise,
                                                            const float* __restrict__ g_map, const float* __restrict__ g_depth,
                                                            const float* __restrict__ g_acc, const float* __restrict__ g_w,
                                                            float* __restrict__ d_raw, float* __restrict__ d_rays_d, int d_rd_stride) {
    const int lane = threadIdx.x & 63;
    const long r = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= R) return;
    const int i0 = lane * SPL;
    const float4* rw = reinterpret_cast<const float4*>(raw + r * (long)S * 4);
    const float* zz = z + r * (long)S;
    float4 v[SPL];
    float zi[SPL + 1], gw[SPL];
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
        const int i = min(i0 + j, S - 1);
        v[j] = rw[i];
        zi[j] = zz[i];
        gw[j] = g_w ? g_w[r * (long)S + i] : 0.f;
    }
    const float* d = rays_d + r * rd_stride;
    const float norm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(d[0], d[0]), __fmul_rn(d[1], d[1])), __fmul_rn(d[2], d[2])));
    const float gm[3] = {g_map ? g_map[r * 3] : 0.f, g_map ? g_map[r * 3 + 1] : 0.f, g_map ? g_map[r * 3 + 2] : 0.f};
    const float gd = g_depth ? g_depth[r] : 0.f;
    const float ga = (g_acc ? g_acc[r] : 0.f) - (white_bkgd ? gm[0] + gm[1] + gm[2] : 0.f);
    zi[SPL] = dpp_f32<0x130>(0.f, zi[0]);
    float alpha[SPL], om[SPL], dist[SPL], spre[SPL], mask[SPL], densm[SPL];
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
        const int i = i0 + j;
        const float vv[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
        dist[j] = 0.f; spre[j] = 0.f; mask[j] = 1.f; densm[j] = 0.f;
        if (i < S - 1) {
            dist[j] = __fmul_rn(__fsub_rn(zi[j + 1], zi[j]), norm);
            spre[j] = vv[sigma_ch] + (noise ? noise[r * (long)(S - 1) + i] : 0.f);
            float dens = act(sigma_act, spre[j]);
            if (rmnear > 0.f) { mask[j] = zi[j + 1] > rmnear ? 1.f : 0.f; dens *= mask[j]; }
            densm[j] = dens;
            alpha[j] = __fadd_rn(-expf(-__fmul_rn(dens, dist[j])), 1.f);
        } else {
            alpha[j] = i == S - 1 ? 1.f : 0.f;
        }
        om[j] = i < S ? __fadd_rn(-alpha[j], 1.f) : 1.f;
    }
    float local = om[0];
#pragma unroll
    for (int j = 1; j < SPL; ++j) local *= om[j];
    const float incl = wave_scan_mul_dpp(local);
    float T = dpp_f32<0x138>(1.f, incl);
    float w[SPL], G[SPL], Tn[SPL], rgbv[SPL][3], lsum = 0.f;
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
        const float vv[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
        w[j] = (i0 + j < S) ? alpha[j] * T : 0.f;
        T *= om[j];
        Tn[j] = T;                                          // T_{i+1}
#pragma unroll
        for (int c = 0; c < 3; ++c) rgbv[j][c] = act(rgb_act, vv[rgb_ch0 + c]);
        G[j] = gm[0] * rgbv[j][0] + gm[1] * rgbv[j][1] + gm[2] * rgbv[j][2] + gd * zi[j] + ga + gw[j];
        lsum += G[j] * w[j];
    }
    const float pre_incl = wave_scan_add_dpp(lsum);          // inclusive prefix of the lanes' sums of G w
    const float total = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pre_incl), 63));
    float run = pre_incl - lsum;                             // exclusive prefix at this lane's first sample
    float dn = 0.f;                                          // this lane's share of |d| dL/d|d|
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
        const int i = i0 + j;
        run += G[j] * w[j];
        if (i < S) {
            const float vv[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
            float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 3; ++c) o[rgb_ch0 + c] = gm[c] * w[j] * act_grad(rgb_act, vv[rgb_ch0 + c]);
            if (i < S - 1) {
                const float suffix = total - run;            // sum_{j > i} G_j w_j
                const float q = dist[j] * (G[j] * Tn[j] - suffix);      // dL / d density_i
                o[sigma_ch] = q * mask[j] * act_grad(sigma_act, spre[j]);
                dn += q * densm[j];
            }
            reinterpret_cast<float4*>(d_raw + r * (long)S * 4)[i] = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    // alpha_i depends on the ray direction through dist_i = dz_i |d| only: dL/d|d| = sum_i (dL/d density_i) density_i / |d|, and
    // d|d| / d d = d / |d|
    if (d_rays_d) {
        dn = wave_sum_dpp(dn);
        if (lane == 0) {
            const float k = dn / (norm * norm);
#pragma unroll
            for (int c = 0; c < 3; ++c) d_rays_d[r * d_rd_stride + c] = k * d[c];
        }
    }
}

// out[r, f] = sum_s w[r, s] * x[r, s, ch0 + f] (optionally through an activation): used for n_rgb != 3 colour
// maps (PBE 15-channel features, voxnerf.py:226) and for feature compositing (nerf.py:119).  One block per ray.
__global__ void k_weighted_channels(const float* __restrict__ x, const float* __restrict__ w, long R, int S, int Cx, int ch0,
                                    int F, int act_code, int add_white, float* __restrict__ out) {
    const long r = blockIdx.x;
    for (int f = threadIdx.x; f < F; f += blockDim.x) {
        float s = 0.f, a = 0.f;
        for (int i = 0; i < S; ++i) {
            const float wi = w[r * (long)S + i];
            s += wi * act(act_code, x[(r * (long)S + i) * Cx + ch0 + f]);
            a += wi;
        }
        out[r * (long)F + f] = add_white ? s + (1.f - a) : s;
    }
}

// ------------------------------------------------------------------------------------------------
// sample_pdf (utils/rays.py:149-193) on bins = z_mid, weights[1:-1] + merge (renderer.py:205,234) + z_std (:250).
// One wavefront per ray; cdf staged in LDS.  The normalising sum and the cdf prefix sums are accumulated in
// float64: for <= 2^29 dynamic range those sums are exact, hence independent of the scan order and bitwise equal
// to torch's CPU cumsum (double accumulator).
__global__ __launch_bounds__(256) void k_sample_pdf_merge(const float* __restrict__ z, const float* __restrict__ wts, long R, int S,
                                                          int N, int det, const float* __restrict__ u,
                                                          float* __restrict__ z_samples, float* __restrict__ z_merged,
                                                          int* __restrict__ order, float* __restrict__ z_std,
                                                          const float* __restrict__ rb, int rb_cols, float* __restrict__ pts_new,
                                                          float* __restrict__ pts_merged) {
    // rb (the ray batch: o at columns 0..2, d at 3..5) + pts_new / pts_merged: the positions o + d z of the new / the merged samples are
    // written here as well (renderer.py:206: the arithmetic of k_points, bit for bit) -- two launches less in a c2f render
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long r = blockIdx.x * (long)(blockDim.x >> 6) + wv;
    const int nb = S - 1;                       // bins (z_mid) = cdf entries
    const int per = 2 * S + N + 2;              // floats of LDS per wave: z[S], cdf[nb]+pad, zs[N]
    float* zs0 = smem_f + (long)wv * per;       // z of the coarse pass
    float* cdf = zs0 + S;                       // [nb]
    float* zsm = cdf + S;                       // new samples [N]
    if (r >= R) return;
    const float* zr = z + r * (long)S;
    const float* wr = wts + r * (long)S;
    float ro[3] = {0.f, 0.f, 0.f}, rd3[3] = {0.f, 0.f, 0.f};
    if (rb) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { ro[c] = rb[r * rb_cols + c]; rd3[c] = rb[r * rb_cols + 3 + c]; }
    }
    for (int i = lane; i < S; i += 64) zs0[i] = zr[i];
    // sum of (w + 1e-5) over weights[1:-1]
    double part = 0.0;
    for (int i = lane; i < nb - 1; i += 64) part += (double)__fadd_rn(wr[i + 1], 1e-5f);
    const float sum = (float)wave_sum_d(part);
    // cdf[0] = 0, cdf[k+1] = float(prefix_double(pdf))
    double carry = 0.0;
    for (int base = 0; base < nb - 1; base += 64) {
        const int i = base + lane;
        const double p = (i < nb - 1) ? (double)(__fadd_rn(wr[i + 1], 1e-5f) / sum) : 0.0;
        const double incl = wave_scan_add_d(p, lane);
        if (i < nb - 1) cdf[i + 1] = (float)(carry + incl);
        carry += __shfl(incl, 63, WAVE);
    }
    if (lane == 0) cdf[0] = 0.f;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    // invert the cdf
    double m1 = 0.0;
    for (int j = lane; j < N; j += 64) {
        const float uu = det ? linspace_at(0.f, 1.f, N, j) : u[r * (long)N + j];
        int lo = 0, hi = nb;                    // first index with cdf[idx] > uu  (searchsorted right=True)
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (cdf[mid] <= uu) lo = mid + 1; else hi = mid; }
        const int below = max(lo - 1, 0), above = min(lo, nb - 1);
        float denom = __fsub_rn(cdf[above], cdf[below]);
        if (denom < 1e-5f) denom = 1.f;
        const float t = __fsub_rn(uu, cdf[below]) / denom;
        const float b0 = __fmul_rn(.5f, __fadd_rn(zs0[below + 1], zs0[below]));   // z_mid on the fly (renderer.py:200)
        const float b1 = __fmul_rn(.5f, __fadd_rn(zs0[above + 1], zs0[above]));
        const float s = __fadd_rn(b0, __fmul_rn(t, __fsub_rn(b1, b0)));
        zsm[j] = s;
        if (z_samples) z_samples[r * (long)N + j] = s;
        if (pts_new) {
#pragma unroll
            for (int c = 0; c < 3; ++c) pts_new[(r * (long)N + j) * 3 + c] = __fadd_rn(ro[c], __fmul_rn(rd3[c], s));
        }
        m1 += (double)s;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (z_std) {                                // torch.std(unbiased=False)
        const double mean = wave_sum_d(m1) / (double)N;
        double v = 0.0;
        for (int j = lane; j < N; j += 64) { const double dd = (double)zsm[j] - mean; v += dd * dd; }
        v = wave_sum_d(v);
        if (lane == 0) z_std[r] = (float)sqrt(v / (double)N);
    }
    if (z_merged || order) {
        // stable rank of every element of cat(z, z_samples): #smaller + #equal-with-lower-index
        const int St = S + N;
        for (int e = lane; e < St; e += 64) {
            const float v = e < S ? zs0[e] : zsm[e - S];
            int rank = 0;
            for (int k = 0; k < S; ++k) { const float o = zs0[k]; rank += (o < v) || (o == v && k < e); }
            for (int k = 0; k < N; ++k) { const float o = zsm[k]; rank += (o < v) || (o == v && (k + S) < e); }
            if (z_merged) z_merged[r * (long)St + rank] = v;
            if (order) order[r * (long)St + rank] = e;
            if (pts_merged) {
#pragma unroll
                for (int c = 0; c < 3; ++c) pts_merged[(r * (long)St + rank) * 3 + c] = __fadd_rn(ro[c], __fmul_rn(rd3[c], v));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// reference networks/renderer.py:259-263: the per-key isnan / isinf guard of render_rays (two host syncs per key there).
// Here: one launch over all keys, flags[k] |= 1 (NaN) | 2 (Inf), no sync; the caller reads the word when it wants to.
constexpr int NUMERICS_MAX_KEYS = 16;
struct NumericsKeys {
    const float* p[NUMERICS_MAX_KEYS];
    long n[NUMERICS_MAX_KEYS];
    int keys;
};
__global__ void k_numerics_flags(const NumericsKeys a, unsigned* __restrict__ flags) {
    const int k = blockIdx.y;
    if (k >= a.keys) return;
    const float* __restrict__ x = a.p[k];
    const long n = a.n[k];
    unsigned f = 0;
    // exponent all ones: mantissa != 0 -> NaN, == 0 -> Inf
    auto look = [&](float v) {
        const unsigned u = __float_as_uint(v);
        if ((u & 0x7f800000u) == 0x7f800000u) f |= (u & 0x007fffffu) ? 1u : 2u;
    };
    const long stride = (long)gridDim.x * blockDim.x;
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x);
        for (long j = i; j < n / 4; j += stride) {
            const float4 v = x4[j];
            look(v.x); look(v.y); look(v.z); look(v.w);
        }
        for (long j = (n / 4) * 4 + i; j < n; j += stride) look(x[j]);
    } else {
        for (long j = i; j < n; j += stride) look(x[j]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) f |= __shfl_xor(f, off, WAVE);
    if ((threadIdx.x & 63) == 0 && f) atomicOr(&flags[k], f);
}


}  // namespace evd
namespace evd {      // internal forms of two entries below, also used by evd_voxel_api.hip (declared in voxel.h)
int launch_sample_z_pts(const evd_render_cfg* cfg, const float* ray_batch, int ncol, long R, const float* t_rand, float* z, float* pts, hipStream_t stream);
}

using namespace evd;

extern "C" {

int evd_get_rays(int H, int W, const float* K, const float* c2w, int add_halfpix, float* rays_o, float* rays_d, void* stream) {
    EVD_REQUIRE(H > 0 && W > 0 && K && c2w && rays_o && rays_d, "evd_get_rays: bad arguments");
    Pose34 pose;
    memcpy(pose.m, c2w, sizeof(pose.m));
    const long n = (long)H * W;
    k_get_rays<<<cdiv(n, 256), 256, 0, as_stream(stream)>>>(H, W, K[0], K[2], K[4], K[5], add_halfpix ? 0.5f : 0.f, pose, rays_o, rays_d);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

int evd_get_rays_pix(const float* coords, const float* K, const float* c2ws, long n, int add_halfpix, float* rays_o, float* rays_d, void* stream) {
    EVD_REQUIRE(n >= 0 && K && rays_o && rays_d, "evd_get_rays_pix: bad arguments");
    if (n == 0) return EVD_OK;
    k_get_rays_pix<<<cdiv(n, 256), 256, 0, as_stream(stream)>>>(coords, K[0], K[2], K[4], K[5], add_halfpix ? 0.5f : 0.f, c2ws, n, rays_o, rays_d);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

static void ndc_coeffs(int H, int W, float focal, float* cw, float* ch) {
    // Python doubles in the reference (utils/rays.py:135-140), rounded to float32 when they meet the tensor
    *cw = (float)(-1.0 / ((double)W / (2.0 * (double)focal)));
    *ch = (float)(-1.0 / ((double)H / (2.0 * (double)focal)));
}

int evd_ndc_rays(int H, int W, float focal, float near, const float* rays_o, const float* rays_d, long n,
                 float* out_o, float* out_d, void* stream) {
    EVD_REQUIRE(n >= 0 && out_o && out_d, "evd_ndc_rays: bad arguments");
    if (n == 0) return EVD_OK;
    float cw, ch;
    ndc_coeffs(H, W, focal, &cw, &ch);
    k_ndc<<<cdiv(n, 256), 256, 0, as_stream(stream)>>>(cw, ch, near, (float)(2.0 * (double)near), rays_o, rays_d, n, out_o, out_d);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

int evd_embed(const float* x, long n, int dim, int L, float* out, void* stream) {
    EVD_REQUIRE(n >= 0 && dim > 0 && L >= 0 && out, "evd_embed: bad arguments");
    if (n == 0) return EVD_OK;
    k_embed<<<cdiv(n * dim, 256), 256, 0, as_stream(stream)>>>(x, n, dim, L, out);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

int evd_rbk_warp(const float* rays, const float* r, const float* v, long R, int M, int use_origin, float* new_rays, float* transforms,
                 void* stream) {
    EVD_REQUIRE(R >= 0 && M >= 1 && (R == 0 || (rays && r && v && new_rays)), "evd_rbk_warp: bad arguments");
    if (R == 0) return EVD_OK;
    const long n = R * (long)(M + (use_origin ? 1 : 0));
    k_rbk_warp<<<cdiv(n, 256), 256, 0, as_stream(stream)>>>(rays, r, v, R, M, use_origin, new_rays, transforms);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

int evd_ray_batch(const evd_render_cfg* cfg, const float* rays, long R, float* ray_batch, void* stream) {
    EVD_REQUIRE(cfg && R >= 0 && ray_batch, "evd_ray_batch: bad arguments");
    if (R == 0) return EVD_OK;
    float cw, ch;
    ndc_coeffs(cfg->H, cfg->W, cfg->focal, &cw, &ch);
    k_ray_batch<<<cdiv(R, 256), 256, 0, as_stream(stream)>>>(rays, R, cfg->ndc, cfg->use_viewdirs, cw, ch, cfg->near, cfg->far, ray_batch);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

int evd_ray_batch_bwd(const evd_render_cfg* cfg, const float* rays, const float* d_ray_batch, long R, float* d_rays, void* stream) {
    EVD_REQUIRE(cfg && rays && d_ray_batch && d_rays && R >= 0 && cfg->use_viewdirs, "evd_ray_batch_bwd: bad arguments (11-column batch)");
    if (R == 0) return EVD_OK;
    float cw, ch;
    ndc_coeffs(cfg->H, cfg->W, cfg->focal, &cw, &ch);
    k_ray_batch_bwd<<<cdiv(R, 256), 256, 0, as_stream(stream)>>>(rays, d_ray_batch, R, cfg->ndc, cw, ch, d_rays);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

int evd_points_bwd(const float* z, const float* d_pts, long R, int S, int accumulate, float* d_ray_batch, void* stream) {
    EVD_REQUIRE(z && d_pts && d_ray_batch && R >= 0 && S >= 1, "evd_points_bwd: bad arguments");
    if (R == 0) return EVD_OK;
    k_points_bwd<<<cdiv(R, 4), 256, 0, as_stream(stream)>>>(z, d_pts, R, S, accumulate, d_ray_batch);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

}  // extern "C"
// internal (evd_api.hip, C++ linkage: not part of the ABI): ray packing + z stratification of evd_nerf_render in one launch
int evd_ray_batch_z(const evd_render_cfg* cfg, const float* rays, long R, const float* t_rand, float* ray_batch, float* z, void* stream) {
    EVD_REQUIRE(cfg && R >= 0 && ray_batch && z && cfg->N_samples > 0 && cfg->use_viewdirs, "evd_ray_batch_z: bad arguments");
    EVD_REQUIRE(!(cfg->perturb > 0.f) || t_rand, "evd_nerf_render: perturb > 0 needs the explicit t_rand draw");
    if (R == 0) return EVD_OK;
    float cw, ch;
    ndc_coeffs(cfg->H, cfg->W, cfg->focal, &cw, &ch);
    k_ray_batch_z<<<cdiv(R * cfg->N_samples, 256), 256, 0, as_stream(stream)>>>(rays, R, cfg->ndc, cw, ch, cfg->near, cfg->far, cfg->N_samples, cfg->lindisp,
                                                                               cfg->perturb > 0.f, t_rand, ray_batch, z);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}
extern "C" {

int evd_sample_z(const evd_render_cfg* cfg, const float* ray_batch, int ncol, long R, const float* t_rand, float* z, void* stream) {
    return evd::launch_sample_z_pts(cfg, ray_batch, ncol, R, t_rand, z, nullptr, as_stream(stream));
}
}  // extern "C"
namespace evd {
// evd_sample_z + the sample positions (internal: evd_c2f_render_rays)
int launch_sample_z_pts(const evd_render_cfg* cfg, const float* ray_batch, int ncol, long R, const float* t_rand, float* z, float* pts, hipStream_t stream) {
    EVD_REQUIRE(cfg && R >= 0 && z && cfg->N_samples > 0, "evd_sample_z: bad arguments");
    EVD_REQUIRE(!(cfg->perturb > 0.f) || t_rand, "evd_sample_z: perturb > 0 needs the explicit t_rand draw");
    EVD_REQUIRE(!pts || ncol >= 6, "evd_sample_z: sample positions need the origin and direction columns");
    if (R == 0) return EVD_OK;
    k_sample_z<<<cdiv(R * cfg->N_samples, 256), 256, 0, stream>>>(ray_batch, ncol, R, cfg->N_samples, cfg->lindisp, cfg->perturb > 0.f, t_rand, z, pts);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}
}  // namespace evd
extern "C" {

int evd_raw2outputs(const float* raw, const float* z, const float* rays_d, int rays_d_stride, long R, int S, int C,
                    int sigma_ch, int rgb_ch0, int n_rgb, int rgb_act, int sigma_act, int white_bkgd,
                    float rmnear_thresh, const float* noise,
                    float* out_map, float* density, float* acc, float* weights, float* depth,
                    const float* feature, int F, float* fmap, void* stream) {
    EVD_REQUIRE(raw && z && rays_d && R >= 0 && S >= 1 && C >= 1, "evd_raw2outputs: bad arguments");
    EVD_REQUIRE(sigma_ch >= 0 && sigma_ch < C && rgb_ch0 >= 0 && rgb_ch0 + n_rgb <= C, "evd_raw2outputs: channel layout out of range");
    EVD_REQUIRE(!(fmap || (n_rgb != 3 && out_map)) || weights, "evd_raw2outputs: weights output required for feature maps");
    if (R == 0) return EVD_OK;
    hipStream_t st = as_stream(stream);
    // interleaved form: the two channel layouts / activation sets the renderer produces (NeRF: rgb sigmoid, sigma last; PDRF levels: sigma
    // first, colour already sigmoided -> relu (coarse) / none (fine)), with the training node's raw noise / density output (not rmnear).  EVD_COMPOSITE_FORM=rows:
    // round 4's kernel; EVD_COMPOSITE_NT=0: plain loads / stores; EVD_COMPOSITE_RPW=1|2|4: rays per wavefront (default by R)
    static const bool il_form = [] { const char* e = getenv("EVD_COMPOSITE_FORM"); return !e || e[0] == 'i'; }();
    static const bool il_nt = [] { const char* e = getenv("EVD_COMPOSITE_NT"); return !e || e[0] != '0'; }();
    static const int il_rpw_env = [] { const char* e = getenv("EVD_COMPOSITE_RPW"); return e ? atoi(e) : 0; }();
    const int il_rpw = il_rpw_env ? il_rpw_env : (R >= (1L << 17) ? 4 : R >= (1L << 14) ? 2 : 1);     // >= 4 workgroups per CU before rays share a wavefront
    if (il_form && n_rgb == 3 && C == 4 && S <= 256 && !(rmnear_thresh > 0.f) && sigma_act == EVD_ACT_RELU &&
        ((sigma_ch == 3 && rgb_ch0 == 0 && rgb_act == EVD_ACT_SIGMOID) ||
         (sigma_ch == 0 && rgb_ch0 == 1 && (rgb_act == EVD_ACT_RELU || rgb_act == EVD_ACT_NONE)))) {
#define EVD_IL3(NCH, RPW, SC, RA, NT) k_composite_il<NCH, RPW, SC, RA, EVD_ACT_RELU, NT><<<(unsigned)cdiv(R, 4 * RPW), 256, 0, st>>>(raw, z, rays_d, rays_d_stride, R, S, white_bkgd, out_map, acc, weights, depth, noise, density)
#define EVD_IL2(NCH, SC, RA, NT) do { if (il_rpw == 4) EVD_IL3(NCH, 4, SC, RA, NT); else if (il_rpw == 1) EVD_IL3(NCH, 1, SC, RA, NT); else EVD_IL3(NCH, 2, SC, RA, NT); } while (0)
#define EVD_IL1(SC, RA, NT) do { if (S <= 64) EVD_IL2(1, SC, RA, NT); else if (S <= 128) EVD_IL2(2, SC, RA, NT); else if (S <= 192) EVD_IL2(3, SC, RA, NT); else EVD_IL2(4, SC, RA, NT); } while (0)
#define EVD_IL0(SC, RA) do { if (il_nt) EVD_IL1(SC, RA, true); else EVD_IL1(SC, RA, false); } while (0)
        if (sigma_ch == 3) EVD_IL0(3, EVD_ACT_SIGMOID);
        else if (rgb_act == EVD_ACT_RELU) EVD_IL0(0, EVD_ACT_RELU);
        else EVD_IL0(0, EVD_ACT_NONE);
#undef EVD_IL0
#undef EVD_IL1
#undef EVD_IL2
#undef EVD_IL3
        EVD_LAUNCH_CHECK();
        if (fmap && feature && F > 0) {
            k_weighted_channels<<<R, F >= 256 ? 256 : 64, 0, st>>>(feature, weights, R, S, F, 0, F, EVD_ACT_NONE, 0, fmap);
            EVD_LAUNCH_CHECK();
        }
        return EVD_OK;
    }
    if (n_rgb == 3 && C == 4 && S <= 256) {
        // bandwidth form: every lane owns SPL consecutive samples
        static const int rpw = [] { const char* e = getenv("EVD_COMPOSITE_RPW"); return e ? atoi(e) : 2; }();
#define EVD_ROWS(SPL) if (rpw == 4) k_composite_rows<SPL, 4><<<cdiv(R, 16), 256, 0, st>>>(raw, z, rays_d, rays_d_stride, R, S, sigma_ch, rgb_ch0, rgb_act, sigma_act, \
                                                                      white_bkgd, rmnear_thresh, noise, out_map, density, acc, weights, depth); \
                      else if (rpw == 2) k_composite_rows<SPL, 2><<<cdiv(R, 8), 256, 0, st>>>(raw, z, rays_d, rays_d_stride, R, S, sigma_ch, rgb_ch0, rgb_act, sigma_act, \
                                                                      white_bkgd, rmnear_thresh, noise, out_map, density, acc, weights, depth); \
                      else k_composite_rows<SPL, 1><<<cdiv(R, 4), 256, 0, st>>>(raw, z, rays_d, rays_d_stride, R, S, sigma_ch, rgb_ch0, rgb_act, sigma_act, \
                                                                      white_bkgd, rmnear_thresh, noise, out_map, density, acc, weights, depth)
        if (S <= 64) EVD_ROWS(1);
        else if (S <= 128) EVD_ROWS(2);
        else if (S <= 192) EVD_ROWS(3);
        else EVD_ROWS(4);
#undef EVD_ROWS
    } else if (n_rgb == 3) {
        k_composite<3><<<cdiv(R, 4), 256, 0, st>>>(raw, z, rays_d, rays_d_stride, R, S, C, sigma_ch, rgb_ch0, n_rgb, rgb_act, sigma_act,
                                                   white_bkgd, rmnear_thresh, noise, out_map, density, acc, weights, depth);
    } else {
        k_composite<0><<<cdiv(R, 4), 256, 0, st>>>(raw, z, rays_d, rays_d_stride, R, S, C, sigma_ch, rgb_ch0, n_rgb, rgb_act, sigma_act,
                                                   white_bkgd, rmnear_thresh, noise, nullptr, density, acc, weights, depth);
        if (out_map && n_rgb > 0)
            k_weighted_channels<<<R, 64, 0, st>>>(raw, weights, R, S, C, rgb_ch0, n_rgb, rgb_act, white_bkgd, out_map);
    }
    EVD_LAUNCH_CHECK();
    if (fmap && feature && F > 0) {
        k_weighted_channels<<<R, F >= 256 ? 256 : 64, 0, st>>>(feature, weights, R, S, F, 0, F, EVD_ACT_NONE, 0, fmap);
        EVD_LAUNCH_CHECK();
    }
    return EVD_OK;
}

int evd_raw2outputs_bwd(const float* raw, const float* z, const float* rays_d, int rays_d_stride, long R, int S, int C,
                        int sigma_ch, int rgb_ch0, int n_rgb, int rgb_act, int sigma_act, int white_bkgd, float rmnear_thresh,
                        const float* noise, const float* g_map, const float* g_depth, const float* g_acc, const float* g_weights,
                        float* d_raw, void* stream) {
    return evd_raw2outputs_bwd_rays(raw, z, rays_d, rays_d_stride, R, S, C, sigma_ch, rgb_ch0, n_rgb, rgb_act, sigma_act, white_bkgd, rmnear_thresh,
                                    noise, g_map, g_depth, g_acc, g_weights, d_raw, nullptr, 0, stream);
}

int evd_raw2outputs_bwd_rays(const float* raw, const float* z, const float* rays_d, int rays_d_stride, long R, int S, int C,
                             int sigma_ch, int rgb_ch0, int n_rgb, int rgb_act, int sigma_act, int white_bkgd, float rmnear_thresh,
                             const float* noise, const float* g_map, const float* g_depth, const float* g_acc, const float* g_weights,
                             float* d_raw, float* d_rays_d, int d_rays_d_stride, void* stream) {
    EVD_REQUIRE(raw && z && rays_d && d_raw && R >= 0 && S >= 1, "evd_raw2outputs_bwd: bad arguments");
    EVD_REQUIRE(!d_rays_d || d_rays_d_stride >= 3, "evd_raw2outputs_bwd_rays: d_rays_d needs a row stride >= 3");
    EVD_REQUIRE(C == 4 && n_rgb == 3 && S <= 256, "evd_raw2outputs_bwd: built for [R,S,4] raw with three colour channels and S <= 256 (got C=%d n_rgb=%d S=%d)", C, n_rgb, S);
    EVD_REQUIRE(sigma_ch >= 0 && sigma_ch < 4 && rgb_ch0 >= 0 && rgb_ch0 + 3 <= 4 && (sigma_ch < rgb_ch0 || sigma_ch >= rgb_ch0 + 3),
                "evd_raw2outputs_bwd: channel layout out of range");
    if (R == 0) return EVD_OK;
    hipStream_t st = as_stream(stream);
#define EVD_BWD(SPL) k_composite_rows_bwd<SPL><<<cdiv(R, 4), 256, 0, st>>>(raw, z, rays_d, rays_d_stride, R, S, sigma_ch, rgb_ch0, rgb_act, sigma_act, \
                                                                          white_bkgd, rmnear_thresh, noise, g_map, g_depth, g_acc, g_weights, d_raw, d_rays_d, d_rays_d_stride)
    if (S <= 64) EVD_BWD(1);
    else if (S <= 128) EVD_BWD(2);
    else if (S <= 192) EVD_BWD(3);
    else EVD_BWD(4);
#undef EVD_BWD
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

}  // extern "C"
namespace evd {
int launch_sample_pdf_merge_pts(const float* z, const float* weights, long R, int S, int N, int det, const float* u,
                                float* z_samples, float* z_merged, int* order, float* z_std,
                                const float* rb, int rb_cols, float* pts_new, float* pts_merged, hipStream_t stream);      // (also declared in voxel.h)
}
extern "C" {
int evd_sample_pdf_merge(const float* z, const float* weights, long R, int S, int N, int det, const float* u,
                         float* z_samples, float* z_merged, int* order, float* z_std, void* stream) {
    return evd::launch_sample_pdf_merge_pts(z, weights, R, S, N, det, u, z_samples, z_merged, order, z_std, nullptr, 0, nullptr, nullptr, as_stream(stream));
}

}  // extern "C"
namespace evd {
// evd_sample_pdf_merge + the positions of the new and of the merged samples (internal: evd_c2f_render_rays)
int launch_sample_pdf_merge_pts(const float* z, const float* weights, long R, int S, int N, int det, const float* u,
                                float* z_samples, float* z_merged, int* order, float* z_std,
                                const float* rb, int rb_cols, float* pts_new, float* pts_merged, hipStream_t stream) {
    EVD_REQUIRE(z && weights && R >= 0 && S >= 3 && N >= 1, "evd_sample_pdf_merge: bad arguments");
    EVD_REQUIRE(rb || (!pts_new && !pts_merged), "evd_sample_pdf_merge: sample positions need the ray batch");
    EVD_REQUIRE(det || u, "evd_sample_pdf_merge: det == 0 needs the explicit u draw");
    if (R == 0) return EVD_OK;
    const size_t lds = 4 * sizeof(float) * (size_t)(2 * S + N + 2);
    EVD_REQUIRE(lds <= 64 * 1024, "evd_sample_pdf_merge: S + N too large for the LDS staging (%zu bytes)", lds);
    k_sample_pdf_merge<<<cdiv(R, 4), 256, lds, stream>>>(z, weights, R, S, N, det, u, z_samples, z_merged, order, z_std, rb, rb_cols, pts_new, pts_merged);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}
}  // namespace evd
extern "C" {

int evd_numerics_flags(const float* const* ptrs, const long* counts, int n_keys, unsigned* flags, void* stream) {
    using namespace evd;
    EVD_REQUIRE(n_keys >= 0 && n_keys <= NUMERICS_MAX_KEYS, "evd_numerics_flags: n_keys %d (max %d)", n_keys, NUMERICS_MAX_KEYS);
    if (n_keys == 0) return EVD_OK;
    EVD_REQUIRE(ptrs && counts && flags, "evd_numerics_flags: null argument");
    NumericsKeys a{};
    a.keys = n_keys;
    long most = 0;
    for (int k = 0; k < n_keys; ++k) {
        EVD_REQUIRE(counts[k] >= 0 && (ptrs[k] || counts[k] == 0), "evd_numerics_flags: key %d is null", k);
        a.p[k] = ptrs[k];
        a.n[k] = counts[k];
        most = counts[k] > most ? counts[k] : most;
    }
    EVD_HIP(hipMemsetAsync(flags, 0, sizeof(unsigned) * n_keys, as_stream(stream)));
    const int bx = (int)std::min<long>(std::max<long>(cdiv(most, 256 * 16), 1), 1024);
    hipLaunchKernelGGL(k_numerics_flags, dim3(bx, n_keys), dim3(256), 0, as_stream(stream), a, flags);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

}  // extern "C"
