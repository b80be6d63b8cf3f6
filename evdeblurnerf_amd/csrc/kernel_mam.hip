// AWP consumer, the per-sample part of the MotionAggregationModule: reference networks/dpnerf/mam.py:72-74 (x_local ->
// self.linear, 64 -> 32 on EVERY sample of every sub-exposure ray) and CorrelationModule.forward :29-33 (a 1x1 attention logit per sample,
// one softmax along the samples and one along the sub-exposures, the two softmax-weighted sums of the `curves`).  As written that is a
// Linear over [R P S, 64], a Conv2d, two softmaxes, two multiplies and two reductions over [R, 32, P, S] tensors -- and the same again
// in the backward.  Because the linear map commutes with the weighted sums (the softmax weights of a row / a column sum to 1) and a
// constant added to every logit leaves a softmax unchanged,
//
//      logit[p,s]         = v . (W h[p,s] + b)         = u . h[p,s] + const,        u = W^T v   (64 floats)
//      curver_inter[:,p]  = sum_s alpha[p,s] (W h[p,s] + b) = W (sum_s alpha[p,s] h[p,s]) + b,   alpha = softmax over s
//      curves_intra[:,s]  = sum_p beta[p,s]  (W h[p,s] + b) = W (sum_p beta[p,s]  h[p,s]) + b,   beta  = softmax over p
//
// so the per-sample work is ONE dot product and two weighted sums of h_local itself; the 64 -> 32 map is applied by the caller to the
// [R, P, 64] and [R, S, 64] results (per-ray sized).  h_local [R P, S, 64] float32 is the AWP embedding's output (awp_embed_kernel.h).
//
// HBM-bound: the forward reads h_local twice (logits, then sums: the softmax needs every logit of the ray first; a ray's 64 P S floats
// -- 320 KiB at P = 10, S = 128 -- do not fit the LDS), the backward reads it once and writes d h_local once.
#include "evd_common.h"
#include "wave_ops.h"

namespace evd {

constexpr int MAM_C = 64, MAM_MAXP = 16, MAM_MAXS = 512;

// sum over the 16 lanes of a DPP row, left in every lane of the row
__device__ __forceinline__ float row_sum16(float v) {
    v += dpp_f32<0xb1>(0.f, v);
    v += dpp_f32<0x4e>(0.f, v);
    v += dpp_f32<0x141>(0.f, v);
    v += dpp_f32<0x140>(0.f, v);
    return v;
}
__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ void fma4(float4& acc, float s, float4 v) {
    acc.x = fmaf(s, v.x, acc.x); acc.y = fmaf(s, v.y, acc.y); acc.z = fmaf(s, v.z, acc.z); acc.w = fmaf(s, v.w, acc.w);
}

// One workgroup per ray; 16 row-groups of 16 lanes, a row-group owns the samples s = g, g + 16, ... of EVERY sub-exposure p (so the
// softmax over p and the intra sum stay inside the group) and a lane 4 of the 64 channels.  A wavefront's four groups read four
// consecutive samples: 1 KiB contiguous per load instruction.
__global__ __launch_bounds__(256) void k_mam_local_fwd(const float* __restrict__ h, const float* __restrict__ u, int P, int S,
                                                       float* __restrict__ h_inter, float* __restrict__ h_intra,
                                                       float* __restrict__ alpha, float* __restrict__ beta) {
    extern __shared__ float lds[];
    float* A = lds;                         // [P][S] logits, then alpha
    float* Bt = A + P * S;                  // [P][S] beta
    float* rmax = Bt + P * S;               // [P] row max, [P] row sum
    float* rsum = rmax + MAM_MAXP;
    float* red = rsum + MAM_MAXP;           // [4 waves][P][64] partial inter sums
    const long b = blockIdx.x;
    const int tid = threadIdx.x, g = tid >> 4, l = tid & 15, wave = tid >> 6, lane = tid & 63;
    const float4* h4 = reinterpret_cast<const float4*>(h) + b * (long)P * S * 16;
    const float4 u4 = reinterpret_cast<const float4*>(u)[l];

    for (int s = g; s < S; s += 16) {
#pragma unroll
        for (int p = 0; p < MAM_MAXP; ++p)
            if (p < P) {
                const float d = row_sum16(dot4(h4[((long)p * S + s) * 16 + l], u4));
                if (l == 0) A[p * S + s] = d;
            }
    }
    __syncthreads();
    // the softmax along the samples: one wavefront per sub-exposure
    for (int p = wave; p < P; p += 4) {
        float m = -INFINITY;
        for (int s = lane; s < S; s += 64) m = fmaxf(m, A[p * S + s]);
        for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float t = 0.f;
        for (int s = lane; s < S; s += 64) t += __expf(A[p * S + s] - m);
        t = wave_sum_dpp(t);
        if (lane == 0) { rmax[p] = m; rsum[p] = t; }
    }
    __syncthreads();
    // the softmax along the sub-exposures: one thread per sample; both weights to LDS and to the saved tensors
    for (int s = tid; s < S; s += 256) {
        float m = -INFINITY, t = 0.f;
        for (int p = 0; p < P; ++p) m = fmaxf(m, A[p * S + s]);
        for (int p = 0; p < P; ++p) t += __expf(A[p * S + s] - m);
        const float it = 1.f / t;
        for (int p = 0; p < P; ++p) {
            const float a = A[p * S + s];
            const float al = __expf(a - rmax[p]) / rsum[p], be = __expf(a - m) * it;
            A[p * S + s] = al;
            Bt[p * S + s] = be;
            alpha[(b * P + p) * S + s] = al;
            beta[(b * P + p) * S + s] = be;
        }
    }
    __syncthreads();
    float4 accP[MAM_MAXP];
#pragma unroll
    for (int p = 0; p < MAM_MAXP; ++p) accP[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = g; s < S; s += 16) {
        float4 accI = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int p = 0; p < MAM_MAXP; ++p)
            if (p < P) {
                const float4 v = h4[((long)p * S + s) * 16 + l];
                fma4(accI, Bt[p * S + s], v);
                fma4(accP[p], A[p * S + s], v);
            }
        reinterpret_cast<float4*>(h_intra)[(b * S + s) * 16 + l] = accI;
    }
    // inter sums: the wavefront's four groups by lane exchange, the four wavefronts through LDS
#pragma unroll
    for (int p = 0; p < MAM_MAXP; ++p)
        if (p < P) {
            float4 a = accP[p];
            a.x += __shfl_xor(a.x, 16); a.y += __shfl_xor(a.y, 16); a.z += __shfl_xor(a.z, 16); a.w += __shfl_xor(a.w, 16);
            a.x += __shfl_xor(a.x, 32); a.y += __shfl_xor(a.y, 32); a.z += __shfl_xor(a.z, 32); a.w += __shfl_xor(a.w, 32);
            if (lane < 16) reinterpret_cast<float4*>(red)[(wave * P + p) * 16 + l] = a;
        }
    __syncthreads();
    for (int i = tid; i < P * MAM_C; i += 256)
        h_inter[b * P * MAM_C + i] = red[i] + red[P * MAM_C + i] + red[2 * P * MAM_C + i] + red[3 * P * MAM_C + i];
}

// The same with the ray's rows REGISTER-RESIDENT between the two phases (P <= PR sub-exposures, S <= 16 NSR samples: 80 sixteen-byte loads
// per thread at the blurfactory shape, all issued in front of the first logit): h_local is read once instead of twice -- the kernel is
// HBM-bound, and a ray's 320 KiB do not fit the LDS but do fit the workgroup's register file (one workgroup of 256 threads per CU).
template <int PR, int NSR>
__global__ __launch_bounds__(256) void k_mam_local_fwd_r(const float* __restrict__ h, const float* __restrict__ u, int P, int S,
                                                         float* __restrict__ h_inter, float* __restrict__ h_intra,
                                                         float* __restrict__ alpha, float* __restrict__ beta) {
    extern __shared__ float lds[];
    float* A = lds;                         // [P][S] logits, then alpha
    float* Bt = A + P * S;                  // [P][S] beta
    float* rmax = Bt + P * S;               // [P] row max, [P] row sum
    float* rsum = rmax + MAM_MAXP;
    float* red = rsum + MAM_MAXP;           // [4 waves][P][64] partial inter sums
    const long b = blockIdx.x;
    const int tid = threadIdx.x, g = tid >> 4, l = tid & 15, wave = tid >> 6, lane = tid & 63;
    const float4* h4 = reinterpret_cast<const float4*>(h) + b * (long)P * S * 16;
    const float4 u4 = reinterpret_cast<const float4*>(u)[l];
    float4 hr[PR][NSR];
#pragma unroll
    for (int p = 0; p < PR; ++p)
#pragma unroll
        for (int j = 0; j < NSR; ++j) {
            const int s = g + 16 * j;
            hr[p][j] = (p < P && s < S) ? h4[((long)p * S + s) * 16 + l] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
    for (int j = 0; j < NSR; ++j)
#pragma unroll
        for (int p = 0; p < PR; ++p) {
            const float d = row_sum16(dot4(hr[p][j], u4));
            if (l == 0 && p < P && g + 16 * j < S) A[p * S + g + 16 * j] = d;
        }
    __syncthreads();
    for (int p = wave; p < P; p += 4) {      // the softmax along the samples: one wavefront per sub-exposure
        float m = -INFINITY;
        for (int s = lane; s < S; s += 64) m = fmaxf(m, A[p * S + s]);
        for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float t = 0.f;
        for (int s = lane; s < S; s += 64) t += __expf(A[p * S + s] - m);
        t = wave_sum_dpp(t);
        if (lane == 0) { rmax[p] = m; rsum[p] = t; }
    }
    __syncthreads();
    for (int s = tid; s < S; s += 256) {     // the softmax along the sub-exposures: one thread per sample
        float m = -INFINITY, t = 0.f;
        for (int p = 0; p < P; ++p) m = fmaxf(m, A[p * S + s]);
        for (int p = 0; p < P; ++p) t += __expf(A[p * S + s] - m);
        const float it = 1.f / t;
        for (int p = 0; p < P; ++p) {
            const float a = A[p * S + s];
            const float al = __expf(a - rmax[p]) / rsum[p], be = __expf(a - m) * it;
            A[p * S + s] = al;
            Bt[p * S + s] = be;
            alpha[(b * P + p) * S + s] = al;
            beta[(b * P + p) * S + s] = be;
        }
    }
    __syncthreads();
    float4 accP[PR];
#pragma unroll
    for (int p = 0; p < PR; ++p) accP[p] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < NSR; ++j) {
        const int s = g + 16 * j;
        if (s < S) {
            float4 accI = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int p = 0; p < PR; ++p)
                if (p < P) {
                    fma4(accI, Bt[p * S + s], hr[p][j]);
                    fma4(accP[p], A[p * S + s], hr[p][j]);
                }
            reinterpret_cast<float4*>(h_intra)[(b * S + s) * 16 + l] = accI;
        }
    }
#pragma unroll
    for (int p = 0; p < PR; ++p)
        if (p < P) {
            float4 a = accP[p];
            a.x += __shfl_xor(a.x, 16); a.y += __shfl_xor(a.y, 16); a.z += __shfl_xor(a.z, 16); a.w += __shfl_xor(a.w, 16);
            a.x += __shfl_xor(a.x, 32); a.y += __shfl_xor(a.y, 32); a.z += __shfl_xor(a.z, 32); a.w += __shfl_xor(a.w, 32);
            if (lane < 16) reinterpret_cast<float4*>(red)[(wave * P + p) * 16 + l] = a;
        }
    __syncthreads();
    for (int i = tid; i < P * MAM_C; i += 256)
        h_inter[b * P * MAM_C + i] = red[i] + red[P * MAM_C + i] + red[2 * P * MAM_C + i] + red[3 * P * MAM_C + i];
}

// Backward.  With gA[p,s] = d_inter[p] . h[p,s], gB[p,s] = d_intra[s] . h[p,s] the two softmax backwards need sum_s alpha gA = d_inter[p] .
// h_inter[p] and sum_p beta gB = d_intra[s] . h_intra[s]: dots of the SAVED outputs, so h_local is read once:
//      d logit = alpha (gA - cP[p]) + beta (gB - cI[s]);   d h = alpha d_inter[p] + beta d_intra[s] + d logit u;   d u = sum d logit h
template <bool ACC>
__global__ __launch_bounds__(256) void k_mam_local_bwd(const float* __restrict__ h, const float* __restrict__ u,
                                                       const float* __restrict__ alpha, const float* __restrict__ beta,
                                                       const float* __restrict__ h_inter, const float* __restrict__ h_intra,
                                                       const float* __restrict__ d_inter, const float* __restrict__ d_intra, int P, int S,
                                                       float* __restrict__ d_h, float* __restrict__ d_u_partial, unsigned* __restrict__ absmax) {
    extern __shared__ float lds[];
    float* dP = lds;                        // [P][64]
    float* cP = dP + MAM_MAXP * MAM_C;      // [P]
    float* red = cP + MAM_MAXP;             // [4][64]
    const long b = blockIdx.x;
    const int tid = threadIdx.x, g = tid >> 4, l = tid & 15, wave = tid >> 6, lane = tid & 63;
    const float4* h4 = reinterpret_cast<const float4*>(h) + b * (long)P * S * 16;
    float4* dh4 = reinterpret_cast<float4*>(d_h) + b * (long)P * S * 16;
    const float4 u4 = reinterpret_cast<const float4*>(u)[l];
    for (int i = tid; i < P * MAM_C; i += 256) dP[i] = d_inter[b * P * MAM_C + i];
    __syncthreads();
    for (int p = g; p < P; p += 16) {
        const float c = row_sum16(dot4(reinterpret_cast<const float4*>(dP)[p * 16 + l],
                                       reinterpret_cast<const float4*>(h_inter)[(b * P + p) * 16 + l]));
        if (l == 0) cP[p] = c;
    }
    __syncthreads();
    float4 du = make_float4(0.f, 0.f, 0.f, 0.f);
    float amax = 0.f;
    for (int s = g; s < S; s += 16) {
        const float4 dI = reinterpret_cast<const float4*>(d_intra)[(b * S + s) * 16 + l];
        const float cI = row_sum16(dot4(dI, reinterpret_cast<const float4*>(h_intra)[(b * S + s) * 16 + l]));
#pragma unroll
        for (int p = 0; p < MAM_MAXP; ++p)
            if (p < P) {
                const long at = ((long)p * S + s) * 16 + l;
                const float4 v = h4[at];
                const float4 dp = reinterpret_cast<const float4*>(dP)[p * 16 + l];
                const float al = alpha[(b * P + p) * S + s], be = beta[(b * P + p) * S + s];
                const float gA = row_sum16(dot4(dp, v)), gB = row_sum16(dot4(dI, v));
                const float da = al * (gA - cP[p]) + be * (gB - cI);
                float4 o;
                o.x = al * dp.x + be * dI.x + da * u4.x;
                o.y = al * dp.y + be * dI.y + da * u4.y;
                o.z = al * dp.z + be * dI.z + da * u4.z;
                o.w = al * dp.w + be * dI.w + da * u4.w;
                if (ACC) {                   // d h_local already holds another consumer's share (the feature integration's): add to it here
                    const float4 prev = dh4[at];  // instead of in a separate pass over 2 x 335 MB
                    o.x += prev.x; o.y += prev.y; o.z += prev.z; o.w += prev.w;
                }
                dh4[at] = o;
                amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
                const float nan_probe = (o.x + o.y) + (o.z + o.w);      // fmaxf drops a NaN: record it as +inf like the scatter's maximum does
                amax = nan_probe != nan_probe ? __builtin_huge_valf() : amax;
                fma4(du, da, v);
            }
    }
    du.x += __shfl_xor(du.x, 16); du.y += __shfl_xor(du.y, 16); du.z += __shfl_xor(du.z, 16); du.w += __shfl_xor(du.w, 16);
    du.x += __shfl_xor(du.x, 32); du.y += __shfl_xor(du.y, 32); du.z += __shfl_xor(du.z, 32); du.w += __shfl_xor(du.w, 32);
    if (lane < 16) reinterpret_cast<float4*>(red)[wave * 16 + l] = du;
    __syncthreads();
    if (tid < MAM_C) d_u_partial[b * MAM_C + tid] = red[tid] + red[MAM_C + tid] + red[2 * MAM_C + tid] + red[3 * MAM_C + tid];
    if (absmax) {                        // max |d h_local| of this ray -> the caller's word (guarded: one atomic per raising workgroup, not per ray)
        for (int o = 32; o; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
        __syncthreads();
        if (lane == 0) red[wave] = amax;
        __syncthreads();
        if (tid == 0) {
            const unsigned mb = __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
            if (mb > *reinterpret_cast<volatile unsigned*>(absmax)) atomicMax(absmax, mb);
        }
    }
}

// ---- h_local's two consumers in ONE backward pass ----------------------------------------------------------------------------------------
// k_awp_integrate_bwd_c64 (kernels_loss.hip) and k_mam_local_bwd above both read h_local [R P, S, 64] and produce a d h_local of that
// size; as two launches the second reads the first one's result back and adds to it (1.67 GB of traffic at the blurfactory shape).  Here
// the MAM backward's 16-lane group that owns row (p, s) also evaluates the integration's gradient of that row -- its formulas need rows
// s - 1 and s + 1 of the same sub-exposure, which the neighbouring groups of the workgroup load anyway (cache hits) -- and writes the sum:
// h_local is fetched from HBM once, d h_local written once.  The integration's d z_vals / d rays_d need per-ray sums of d dist: the rows'
// values go through LDS ([P][S]) and are folded after the tile loop.  Same arithmetic as the two kernels (awp.py:58-75 as written:
// Q[s] is the cumulative product over the CHANNELS of row s - 1's 1 - alpha + 1e-10).
struct LocalBwdParams {
    const float *h, *u, *alpha, *beta, *h_inter, *h_intra, *d_inter, *d_intra;     // the MAM part (k_mam_local_bwd)
    const float *z, *rays_d, *d_int;                                                 // the integration part: z [R P, S], rays_d [R P, 3], d out [R P, 64]
    int P, S;
    float *d_h, *d_u_partial, *d_z, *d_rays_d;                                       // d_z, d_rays_d: may be null
    unsigned* absmax;
};
__device__ __forceinline__ float lc_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
template <int CTRL> __device__ __forceinline__ float lc_dpp(float old, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float lc_scan_mul(float v) {          // inclusive product over the lanes <= this one of the 16-lane row
    v *= lc_dpp<0x111>(1.f, v); v *= lc_dpp<0x112>(1.f, v); v *= lc_dpp<0x114>(1.f, v); v *= lc_dpp<0x118>(1.f, v);
    return v;
}
__device__ __forceinline__ float lc_scan_add_right(float v) {    // inclusive sum over the lanes >= this one
    v += lc_dpp<0x101>(0.f, v); v += lc_dpp<0x102>(0.f, v); v += lc_dpp<0x104>(0.f, v); v += lc_dpp<0x108>(0.f, v);
    return v;
}

__global__ __launch_bounds__(256) void k_local_consumers_bwd(const LocalBwdParams q) {
    extern __shared__ float lds[];
    const int P = q.P, S = q.S;
    float* dP = lds;                        // [MAXP][64]
    float* cP = dP + MAM_MAXP * MAM_C;      // [MAXP]
    float* red = cP + MAM_MAXP;             // [4][64]
    float* nrm = red + 4 * MAM_C;           // [MAXP] |rays_d| of the sub-exposures
    float* zl = nrm + MAM_MAXP;             // [P][S] z_vals of the ray's sub-exposures
    float* dd = zl + P * S;                 // [P][S] d dist of every row
    const long b = blockIdx.x;
    const int tid = threadIdx.x, g = tid >> 4, l = tid & 15, wave = tid >> 6, lane = tid & 63;
    const float4* h4 = reinterpret_cast<const float4*>(q.h) + b * (long)P * S * 16;
    float4* dh4 = reinterpret_cast<float4*>(q.d_h) + b * (long)P * S * 16;
    const float4 u4 = reinterpret_cast<const float4*>(q.u)[l];
    for (int i = tid; i < P * MAM_C; i += 256) dP[i] = q.d_inter[b * P * MAM_C + i];
    for (int i = tid; i < P * S; i += 256) zl[i] = q.z[b * P * S + i];
    if (tid < P) {
        const float* d = q.rays_d + (b * P + tid) * 3;
        nrm[tid] = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(d[0], d[0]), __fmul_rn(d[1], d[1])), __fmul_rn(d[2], d[2])));
    }
    __syncthreads();
    for (int p = g; p < P; p += 16) {
        const float c = row_sum16(dot4(reinterpret_cast<const float4*>(dP)[p * 16 + l],
                                       reinterpret_cast<const float4*>(q.h_inter)[(b * P + p) * 16 + l]));
        if (l == 0) cP[p] = c;
    }
    __syncthreads();
    float4 du = make_float4(0.f, 0.f, 0.f, 0.f);
    float amax = 0.f;
    for (int s = g; s < S; s += 16) {
        const float4 dI = reinterpret_cast<const float4*>(q.d_intra)[(b * S + s) * 16 + l];
        const float cI = row_sum16(dot4(dI, reinterpret_cast<const float4*>(q.h_intra)[(b * S + s) * 16 + l]));
        const bool last = s == S - 1;
#pragma unroll 2
        for (int p = 0; p < P; ++p) {
            const long at = ((long)p * S + s) * 16 + l;
            const float4 v = h4[at];
            const float4 vm = h4[s > 0 ? at - 16 : at], vn = h4[last ? at : at + 16];
            const float4 gi = reinterpret_cast<const float4*>(q.d_int)[(b * P + p) * 16 + l];
            const float4 dp = reinterpret_cast<const float4*>(dP)[p * 16 + l];
            const float al = q.alpha[(b * P + p) * S + s], be = q.beta[(b * P + p) * S + s];
            // ---- the MAM's share (k_mam_local_bwd)
            const float gA = row_sum16(dot4(dp, v)), gB = row_sum16(dot4(dI, v));
            const float da = al * (gA - cP[p]) + be * (gB - cI);
            float o[4] = {al * dp.x + be * dI.x + da * u4.x, al * dp.y + be * dI.y + da * u4.y, al * dp.z + be * dI.z + da * u4.z,
                          al * dp.w + be * dI.w + da * u4.w};
            fma4(du, da, v);
            // ---- the integration's share (k_awp_integrate_bwd_c64, row s of sub-exposure p)
            const float* zz = zl + p * S;
            const float norm = nrm[p];
            const float dz = last ? 0.f : __fsub_rn(zz[s + 1], zz[s]);
            const float dist = __fmul_rn(dz, norm);
            const float dist_m = s > 0 ? __fmul_rn(__fsub_rn(zz[s], zz[s - 1]), norm) : 0.f;
            const float dist_n = s + 2 < S ? __fmul_rn(__fsub_rn(zz[s + 2], zz[s + 1]), norm) : 0.f;
            const float fc[4] = {v.x, v.y, v.z, v.w}, fm[4] = {vm.x, vm.y, vm.z, vm.w}, fn[4] = {vn.x, vn.y, vn.z, vn.w}, gg[4] = {gi.x, gi.y, gi.z, gi.w};
            float Q[4], e[4], om[4], Qn[4], G[4], lsum = 0.f, localm = 1.f, local = 1.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                Q[k] = s > 0 ? __fadd_rn(lc_exp(-__fmul_rn(fm[k], dist_m)), 1e-10f) : 1.f;      // om of row s - 1 (never the last row)
                localm *= Q[k];
                e[k] = last ? 1.f : lc_exp(-__fmul_rn(fc[k], dist));
                om[k] = __fadd_rn(e[k], 1e-10f);
                local *= om[k];
            }
            float exm = lc_dpp<0x111>(1.f, lc_scan_mul(localm)), ex = lc_dpp<0x111>(1.f, lc_scan_mul(local));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                exm *= Q[k]; Q[k] = s > 0 ? exm : 1.f;            // Q of this row: the cumulative product of row s - 1's om over the channels
                ex *= om[k]; Qn[k] = ex;
                const float en = s + 2 < S ? lc_exp(-__fmul_rn(fn[k], dist_n)) : 1.f;
                const float an = s + 2 < S ? __fadd_rn(-en, 1.f) : 0.f;
                G[k] = last ? 0.f : gg[k] * an * fn[k] * Qn[k];
                lsum += G[k];
            }
            float sfx = lc_scan_add_right(lsum) - lsum, ddist = 0.f;
#pragma unroll
            for (int k = 3; k >= 0; --k) {
                sfx += G[k];
                const float a = last ? 0.f : __fadd_rn(-e[k], 1.f);
                const float through = last ? 0.f : sfx * __builtin_amdgcn_rcpf(om[k]);
                const float ga = gg[k] * Q[k] * fc[k] - through;
                o[k] += last ? 0.f : gg[k] * Q[k] * a + ga * dist * e[k];
                ddist += last ? 0.f : ga * fc[k] * e[k];
            }
            ddist = row_sum16(ddist);
            if (l == 0) dd[p * S + s] = ddist;
            dh4[at] = make_float4(o[0], o[1], o[2], o[3]);
            amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
            const float nan_probe = (o[0] + o[1]) + (o[2] + o[3]);      // a NaN (or inf - inf) in d h_local: +inf, not a finite scale word
            amax = nan_probe != nan_probe ? __builtin_huge_valf() : amax;
        }
    }
    du.x += __shfl_xor(du.x, 16); du.y += __shfl_xor(du.y, 16); du.z += __shfl_xor(du.z, 16); du.w += __shfl_xor(du.w, 16);
    du.x += __shfl_xor(du.x, 32); du.y += __shfl_xor(du.y, 32); du.z += __shfl_xor(du.z, 32); du.w += __shfl_xor(du.w, 32);
    if (lane < 16) reinterpret_cast<float4*>(red)[wave * 16 + l] = du;
    __syncthreads();
    if (tid < MAM_C) q.d_u_partial[b * MAM_C + tid] = red[tid] + red[MAM_C + tid] + red[2 * MAM_C + tid] + red[3 * MAM_C + tid];
    // d z[s] = (d dist[s - 1] - d dist[s]) |d|;  d rays_d = (sum_s d dist[s] (z[s + 1] - z[s])) d / |d|      (awp.py:61-63)
    if (q.d_z)
        for (int i = tid; i < P * S; i += 256) {
            const int p = i / S, sx = i - p * S;
            q.d_z[b * P * S + i] = ((sx > 0 ? dd[i - 1] : 0.f) - dd[i]) * nrm[p];
        }
    if (q.d_rays_d)
        for (int p = wave; p < P; p += 4) {
            float a = 0.f;
            for (int sx = lane; sx < S - 1; sx += 64) a = fmaf(dd[p * S + sx], zl[p * S + sx + 1] - zl[p * S + sx], a);
            a = wave_sum_dpp(a);
            if (lane < 3) {
                const float* d = q.rays_d + (b * P + p) * 3;
                q.d_rays_d[(b * P + p) * 3 + lane] = nrm[p] > 0.f ? a * d[lane] / nrm[p] : 0.f;
            }
        }
    if (q.absmax) {
        for (int o = 32; o; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
        __syncthreads();
        if (lane == 0) red[wave] = amax;
        __syncthreads();
        if (tid == 0) {
            const unsigned mb = __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
            if (mb > *reinterpret_cast<volatile unsigned*>(q.absmax)) atomicMax(q.absmax, mb);
        }
    }
}

}  // namespace evd

using namespace evd;

static int mam_check(const char* who, long R, int P, int S, int C) {
    EVD_REQUIRE(R >= 0 && P >= 1 && S >= 1, "%s: bad sizes", who);
    EVD_REQUIRE(C == MAM_C, "%s: %d channels (built: the embedding's %d)", who, C, MAM_C);
    EVD_REQUIRE(P <= MAM_MAXP && S <= MAM_MAXS, "%s: P = %d, S = %d (built: P <= %d, S <= %d)", who, P, S, MAM_MAXP, MAM_MAXS);
    return EVD_OK;
}

int evd_mam_local_forward(const float* h_local, const float* u, long R, int P, int S, int C, float* h_inter, float* h_intra, float* alpha,
                          float* beta, void* stream) {
    EVD_REQUIRE(h_local && u && h_inter && h_intra && alpha && beta, "evd_mam_local_forward: null argument");
    if (int e = mam_check("evd_mam_local_forward", R, P, S, C)) return e;
    if (R == 0) return EVD_OK;
    const size_t lds = sizeof(float) * ((size_t)2 * P * S + 2 * MAM_MAXP + (size_t)4 * P * MAM_C);
    constexpr size_t lds_max = sizeof(float) * ((size_t)2 * MAM_MAXP * MAM_MAXS + 2 * MAM_MAXP + (size_t)4 * MAM_MAXP * MAM_C);
    EVD_SET_MAX_LDS(k_mam_local_fwd, lds_max);
    static const bool two_pass = getenv("EVD_MAM_TWO_PASS") != nullptr;       // developer switch: round 3's kernel
    if (!two_pass && P <= 10 && S <= 128) {
        EVD_SET_MAX_LDS((k_mam_local_fwd_r<10, 8>), lds_max);
        k_mam_local_fwd_r<10, 8><<<(unsigned)R, 256, lds, as_stream(stream)>>>(h_local, u, P, S, h_inter, h_intra, alpha, beta);
    } else
        k_mam_local_fwd<<<(unsigned)R, 256, lds, as_stream(stream)>>>(h_local, u, P, S, h_inter, h_intra, alpha, beta);
    EVD_HIP(hipGetLastError());
    return EVD_OK;
}

int evd_mam_local_backward(const float* h_local, const float* u, const float* alpha, const float* beta, const float* h_inter,
                           const float* h_intra, const float* d_inter, const float* d_intra, long R, int P, int S, int C, float* d_h_local,
                           float* d_u_partial, int accumulate, unsigned* d_h_absmax, void* stream) {
    EVD_REQUIRE(h_local && u && alpha && beta && h_inter && h_intra && d_inter && d_intra && d_h_local && d_u_partial,
                "evd_mam_local_backward: null argument");
    if (int e = mam_check("evd_mam_local_backward", R, P, S, C)) return e;
    if (R == 0) return EVD_OK;
    if (d_h_absmax) EVD_HIP(hipMemsetAsync(d_h_absmax, 0, sizeof(unsigned), as_stream(stream)));     // the kernel raises the word; the entry owns its start value
    const size_t lds = sizeof(float) * (MAM_MAXP * MAM_C + MAM_MAXP + 4 * MAM_C);
    if (accumulate)
        k_mam_local_bwd<true><<<(unsigned)R, 256, lds, as_stream(stream)>>>(h_local, u, alpha, beta, h_inter, h_intra, d_inter, d_intra, P, S,
                                                                            d_h_local, d_u_partial, d_h_absmax);
    else
        k_mam_local_bwd<false><<<(unsigned)R, 256, lds, as_stream(stream)>>>(h_local, u, alpha, beta, h_inter, h_intra, d_inter, d_intra, P, S,
                                                                             d_h_local, d_u_partial, d_h_absmax);
    EVD_HIP(hipGetLastError());
    return EVD_OK;
}

int evd_awp_local_consumers_backward(const float* h_local, const float* u, const float* alpha, const float* beta, const float* h_inter,
                                     const float* h_intra, const float* d_inter, const float* d_intra, const float* z, const float* rays_d,
                                     const float* d_integrated, long R, int P, int S, int C, float* d_h_local, float* d_u_partial, float* d_z,
                                     float* d_rays_d, unsigned* d_h_absmax, void* stream) {
    EVD_REQUIRE(h_local && u && alpha && beta && h_inter && h_intra && d_inter && d_intra && z && rays_d && d_integrated && d_h_local && d_u_partial,
                "evd_awp_local_consumers_backward: null argument");
    if (int e = mam_check("evd_awp_local_consumers_backward", R, P, S, C)) return e;
    if (R == 0) return EVD_OK;
    if (d_h_absmax) EVD_HIP(hipMemsetAsync(d_h_absmax, 0, sizeof(unsigned), as_stream(stream)));
    LocalBwdParams q{};
    q.h = h_local; q.u = u; q.alpha = alpha; q.beta = beta; q.h_inter = h_inter; q.h_intra = h_intra; q.d_inter = d_inter; q.d_intra = d_intra;
    q.z = z; q.rays_d = rays_d; q.d_int = d_integrated; q.P = P; q.S = S;
    q.d_h = d_h_local; q.d_u_partial = d_u_partial; q.d_z = d_z; q.d_rays_d = d_rays_d; q.absmax = d_h_absmax;
    const size_t lds = sizeof(float) * (MAM_MAXP * MAM_C + MAM_MAXP + 4 * MAM_C + MAM_MAXP + (size_t)2 * P * S);
    EVD_SET_MAX_LDS(k_local_consumers_bwd, sizeof(float) * (MAM_MAXP * MAM_C + 2 * MAM_MAXP + 4 * MAM_C + (size_t)2 * MAM_MAXP * MAM_MAXS));
    k_local_consumers_bwd<<<(unsigned)R, 256, lds, as_stream(stream)>>>(q);
    EVD_HIP(hipGetLastError());
    return EVD_OK;
}
