// AWP consumer, the PER-RAY remainder of the adaptive weight proposal: reference networks/dpnerf/awp.py:89-95 (direction
// encoding), :104-109 (motion_feature_embed_layer), networks/dpnerf/mam.py:35-53 (CorrelationModule.forward behind its per-sample part)
// and awp.py:112-115 (mean over the sub-exposures, w_linear, sigmoid, normalisation).  The reference runs ~100 launches of
// [R, 32, P]- and [R, 32, S]-sized tensors each way; here a workgroup owns a ray, keeps every intermediate of that ray in LDS
// (P <= 16 sub-exposures x 32 channels, S samples x 32 / 16 channels) and walks the whole chain:
//
//   forward   k_awp_tail_fwd     per ray: x_global, attention over P and over S, convd -> y [R,P,32]; per-workgroup sums of y, y^2 (double)
//             k_awp_tail_finish  folds the sums into the BatchNorm statistics (over ALL rays), running estimates, then per ray:
//                                normalise, residual, leaky_relu, mean over P, w_linear, sigmoid, normalise -> out [R,P]
//   backward  k_awp_tail_bwd0    per ray: d out -> d z [R,P,32]; partial sums of d gamma, d beta, d w_linear
//             k_awp_tail_bwd     per ray: BatchNorm backward with the folded sums, the forward recomputed in LDS, the chain backwards;
//                                input gradients to HBM, parameter gradients added into the workgroup's partial row
//             k_awp_tail_reduce  the partial rows -> d_params
//
// float32 on the vector ALU throughout (0.45 M multiply-adds per ray forward: the matrix cores have nothing to win at these shapes);
// the per-sample-sized products (MAM.linear / convb / convl on [S, 64 -> 32 -> 16 -> 16]) run one sample row per lane pair with the row
// in registers and the weights as LDS broadcasts; everything P-sized goes through one strided helper.
#include "evd_common.h"
#include "wave_ops.h"

namespace evd {

constexpr int AT_WS = 64, AT_CM = 32, AT_MID = 16, AT_MAXP = 16, AT_MAXMOT = 4, AT_MAXW = 2 * AT_MAXMOT + 12;
// threads of a ray's workgroup, and how many of them share a sample row in the per-sample phases.  The working set of a ray allows one
// workgroup per CU, so the wavefronts that hide each other's LDS latency all come from here: 256 threads (one wavefront per SIMD) measured
// 66 k cycles per ray forward, 207 k backward
constexpr int AT_NT = 512, AT_LPS = 4, AT_O64 = AT_WS / AT_LPS, AT_O32 = AT_CM / AT_LPS, AT_O16 = AT_MID / AT_LPS;
constexpr int AT_LS = AT_CM + 1, AT_LK = AT_MID + 1;      // padded row strides of the per-sample LDS arrays (conflict-free row writes)

struct TailDims {
    int P, S, VF, F, VC, IN0, n_mot, SA;                  // VC = view columns (VF + direction encoding), IN0 = 64 + VC, SA = S + 1
};

__host__ __device__ inline TailDims tail_dims(int P, int S, int VF, int F, int n_mot) {
    TailDims d;
    d.P = P; d.S = S; d.VF = VF; d.F = F; d.n_mot = n_mot;
    d.VC = VF + (F >= 0 ? 3 + 6 * F : 0);
    d.IN0 = AT_WS + d.VC;
    d.SA = S + 1;
    return d;
}

// parameter tensors in API order; sizes in floats
__host__ __device__ inline int tail_param_size(const TailDims& d, int i) {
    if (i < 2 * d.n_mot) return (i & 1) ? AT_CM : AT_CM * (i == 0 ? d.IN0 : AT_CM);
    switch (i - 2 * d.n_mot) {
        case 0: return AT_CM * AT_WS;      // MAM.linear.weight
        case 1: return AT_CM;              // MAM.linear.bias
        case 2: case 3: case 4: return AT_MID * AT_CM;      // conva, convb, convc
        case 5: case 6: return AT_MID * AT_MID;             // convn, convl
        case 7: return AT_CM * AT_CM;      // convd.0
        case 8: case 9: return AT_CM;      // convd.1 weight, bias
        case 10: return d.P * AT_CM;       // w_linear.weight
        default: return d.P;               // w_linear.bias
    }
}
enum { TW_LIN_W = 0, TW_LIN_B, TW_CONVA, TW_CONVB, TW_CONVC, TW_CONVN, TW_CONVL, TW_CONVD, TW_BN_W, TW_BN_B, TW_WL_W, TW_WL_B, TW_COUNT };

struct TailLds {
    float *mw0, *mb0, *mwr, *lin_w, *lin_b, *conva, *convb, *convc, *convn, *convl, *convd;      // mwr: layers 1.. as (W [32,32], b [32]) records
    float *x0, *xs0, *hi, *li, *kP, *nP, *q, *aP, *f, *yb;                                         // xs0: the layers' outputs, [n_mot][P][32]
    int xs_stride, ray_floats;                 // ray_floats: x0 .. aS, the ray's forward state (one contiguous block)
    __host__ __device__ float* mw(int l) const { return l == 0 ? mw0 : mwr + (l - 1) * (AT_CM * AT_CM + AT_CM); }
    __host__ __device__ float* mb(int l) const { return l == 0 ? mb0 : mwr + (l - 1) * (AT_CM * AT_CM + AT_CM) + AT_CM * AT_CM; }
    __host__ __device__ float* xs(int l) const { return xs0 + l * xs_stride; }
    float *ls, *kI, *nI, *aS;
    float *df, *daP, *dnP, *dkP, *dq, *dli, *dxa, *dxb, *daS, *dkI, *dls, *tmp;
};

// one layout for the host (size) and the device (pointers); every array starts 16-byte aligned
__host__ __device__ inline size_t tail_lds_layout(float* base, const TailDims& d, bool bwd, TailLds& L) {
    size_t at = 0;
    auto take = [&](size_t n) { float* p = base + at; at += (n + 3) & ~(size_t)3; return p; };
    L.mw0 = take((size_t)AT_CM * d.IN0); L.mb0 = take(AT_CM);
    L.mwr = take((size_t)(d.n_mot - 1) * (AT_CM * AT_CM + AT_CM));
    L.lin_w = take(AT_CM * AT_WS); L.lin_b = take(AT_CM);
    L.conva = take(AT_MID * AT_CM); L.convb = take(AT_MID * AT_CM); L.convc = take(AT_MID * AT_CM);
    L.convn = take(AT_MID * AT_MID); L.convl = take(AT_MID * AT_MID); L.convd = take(AT_CM * AT_CM);
    const size_t P = d.P;
    L.x0 = take(P * d.IN0);
    L.xs_stride = (int)(P * AT_CM);
    L.xs0 = take((size_t)d.n_mot * P * AT_CM);
    L.hi = take(P * AT_WS); L.li = take(P * AT_CM); L.kP = take(P * AT_MID); L.nP = take(P * AT_MID); L.q = take(P * AT_MID);
    L.aP = take(P * P); L.f = take(P * AT_CM); L.yb = take(P * AT_CM);
    L.ls = take((size_t)d.S * AT_LS); L.kI = take((size_t)d.S * AT_LK); L.nI = take((size_t)d.S * AT_LK); L.aS = take(P * d.SA);
    L.ray_floats = (int)(base + at - L.x0);
    L.df = L.daP = L.dnP = L.dkP = L.dq = L.dli = L.dxa = L.dxb = L.daS = L.dkI = L.dls = L.tmp = nullptr;
    if (bwd) {
        const size_t wide = d.IN0 > AT_WS ? d.IN0 : AT_WS;
        L.df = take(P * AT_CM); L.daP = take(P * P); L.dnP = take(P * AT_MID); L.dkP = take(P * AT_MID); L.dq = take(P * AT_MID);
        L.dli = take(P * AT_CM); L.dxa = take(P * wide); L.dxb = take(P * wide);
        L.daS = take(P * d.SA); L.dkI = take((size_t)d.S * AT_LK); L.dls = take((size_t)d.S * AT_LS); L.tmp = take(64);
    }
    return at;
}

struct TailKParams {
    const float* w[AT_MAXW];
    long off[AT_MAXW + 1];             // offsets of the tensors in the flat gradient buffer; [nw] = total
    TailDims d;
    int training, nw;
    float eps, momentum;
    long R;
    const float *h, *vf, *rays_d, *h_inter, *h_intra;
    float *y, *xg;                     // [R, P, 32] (forward: written; backward: read)
    float* saved;                      // [R, ray_floats]: every ray's forward state (NULL: the backward recomputes it)
    double* bn_part;                   // forward: [grid][64] sums of y, y^2
    const float *dz, *stats, *partB;   // backward
    int nblkB, partB_stride;
    long partA_stride;                 // floats between the workgroups' partial rows (a multiple of 4: the rows are 16-byte aligned)
    float *d_h, *d_vf, *d_rays, *d_hi, *d_hs, *partA;
    long long* stamps;                 // developer build (-DEVD_AT_STAMP): per-phase shader-clock sums of workgroup 0
};

// developer build (-DEVD_AT_STAMP, tools/dev/stamp_awp_tail.py): thread 0 of workgroup 0 sums the shader-clock cycles between marks
struct AtClock {
#ifdef EVD_AT_STAMP
    long long t[32], last;
    __device__ __forceinline__ void start() {
        for (int i = 0; i < 32; ++i) t[i] = 0;
        last = __builtin_readcyclecounter();
    }
    __device__ __forceinline__ void mark(int i) {
        const long long now = __builtin_readcyclecounter();
        t[i] += now - last;
        last = now;
    }
    __device__ __forceinline__ void flush(long long* out) const {
        if (out && blockIdx.x == 0 && threadIdx.x == 0)
            for (int i = 0; i < 32; ++i) out[i] = t[i];
    }
#else
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void mark(int) {}
    __device__ __forceinline__ void flush(long long*) const {}
#endif
};

// threadIdx.x behind an opaque barrier: hipcc otherwise hoists every helper's per-thread index arithmetic (i / N, n % K, the strided
// offsets of ~40 calls) out of the ray loop and spills it (measured: 278 / 368 spilled registers at 512 threads)
__device__ __forceinline__ int at_tid() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

// The sample-sized products on the exact-float32 matrix core (v_mfma_f32_16x16x4_f32 = an fmaf chain in k order): one wavefront, one
// 16 x 16 tile  D[m][n] += sum_k A[m sa_m + k sa_k] B[k sb_k + n sb_n],  m < Mv, n < Nv, k < K (operands outside are zeros); four
// k-steps of operands are loaded in front of their four MFMAs.  Lane l holds A[m = l % 16][k = l / 16], B[k = l / 16][n = l % 16] and
// the results D[4 (l / 16) + i][l % 16] in acc[i].  On the vector ALU these were the lane-serial loops over the 128 samples of a ray:
// the attention sums (8 k cycles of a 64 k-cycle forward), MAM.linear on the intra sums (28 k), its two transposes in the backward (35 k).
typedef float f32x4v __attribute__((ext_vector_type(4)));
static_assert(AT_NT == 512, "the MAM.linear weight gradient is eight 16 x 16 tiles, one per wavefront");
__device__ __forceinline__ f32x4v tile16(f32x4v acc, const float* A, int sa_m, int sa_k, int Mv, const float* B, int sb_k, int sb_n, int Nv,
                                         int K) {
    const int lane = at_tid() & 63, mn = lane & 15, kq = lane >> 4;
    const bool am = mn < Mv, bn = mn < Nv;
    const float* a = A + (am ? mn : 0) * sa_m;
    const float* b = B + (bn ? mn : 0) * sb_n;
    int k0 = 0;
    for (; k0 + 32 <= K; k0 += 32) {                       // eight k-steps of operands in flight (a trip costs one LDS round trip, whatever its width)
        float av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + 4 * u + kq;
            av[u] = a[k * sa_k];
            bv[u] = b[k * sb_k];
            av[u] = am ? av[u] : 0.f;
            bv[u] = bn ? bv[u] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
    }
    for (; k0 < K; k0 += 16) {
        float av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + 4 * u + kq;
            const bool in = k < K;
            const int kc = in ? k : 0;
            av[u] = a[kc * sa_k];
            bv[u] = b[kc * sb_k];
            av[u] = (am && in) ? av[u] : 0.f;
            bv[u] = (bn && in) ? bv[u] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
    }
    return acc;
}
__device__ __forceinline__ f32x4v zero4() {
    f32x4v z;
    z[0] = z[1] = z[2] = z[3] = 0.f;
    return z;
}

// Every product of the chain is a handful of 16 x 16 tiles; a tile goes to the wavefront whose number the running job counter names (the
// same sequence in every thread and for every ray: a parameter's gradient tile is always added by the same lanes).
struct TileJobs {
    int next;
    __device__ __forceinline__ bool mine() { return ((next++) & (AT_NT / 64 - 1)) == (at_tid() >> 6); }
};
// out[m som + n] (+)= act(bias[n] + sum_k A[m sam + k sak] B[n sbn + k sbk]),  m < M <= 16
template <int ACT, bool ACC>
__device__ __forceinline__ void mm(TileJobs& jobs, float* out, int som, const float* A, int sam, int sak, const float* B, int sbn, int sbk,
                                   const float* bias, int M, int N, int K) {
    const int lane = at_tid() & 63, tr = 4 * (lane >> 4), tc = lane & 15;
    for (int n0 = 0; n0 < N; n0 += 16) {
        if (!jobs.mine()) continue;
        const f32x4v acc = tile16(zero4(), A, sam, sak, M, B + n0 * sbn, sbk, sbn, min(16, N - n0), K);
        if (n0 + tc < N) {
            const float bv = bias ? bias[n0 + tc] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (tr + i < M) {
                    float v = acc[i] + bv;
                    if (ACT == 1) v = fmaxf(v, 0.f);
                    if (ACC) out[(tr + i) * som + n0 + tc] += v;
                    else out[(tr + i) * som + n0 + tc] = v;
                }
        }
    }
}
// y = x W^T (+ b) with W [N, K] row-major (nn.Linear / 1x1 convolution)
template <int ACT>
__device__ __forceinline__ void lin(TileJobs& jobs, float* out, int som, const float* x, int ldx, const float* W, const float* b, int M, int N, int K) {
    mm<ACT, false>(jobs, out, som, x, ldx, 1, W, K, 1, b, M, N, K);
}
// part[n K + k] += sum_m G[m sgm + n] X[m sxm + k]  (weight gradient of y = x W^T)
__device__ __forceinline__ void wacc(TileJobs& jobs, float* part, const float* G, int sgm, const float* X, int sxm, int M, int N, int K) {
    const int lane = at_tid() & 63, tr = 4 * (lane >> 4), tc = lane & 15;
    for (int n0 = 0; n0 < N; n0 += 16)
        for (int k0 = 0; k0 < K; k0 += 16) {
            if (!jobs.mine()) continue;
            float old[4];                                   // (the partial row lives in L2: its read is in flight while the tile is computed)
#pragma unroll
            for (int i = 0; i < 4; ++i) old[i] = (k0 + tc < K && n0 + tr + i < N) ? part[(n0 + tr + i) * K + k0 + tc] : 0.f;
            const f32x4v acc = tile16(zero4(), G + n0, 1, sgm, min(16, N - n0), X + k0, sxm, 1, min(16, K - k0), M);
            if (k0 + tc < K) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (n0 + tr + i < N) part[(n0 + tr + i) * K + k0 + tc] = old[i] + acc[i];
            }
        }
}
// part[n] += sum_m G[m sgm + n]: the same tile against a column of ones (`one` points at a 1.0f in LDS)
__device__ __forceinline__ void bacc(TileJobs& jobs, float* part, const float* G, int sgm, int M, int N, const float* one) {
    const int lane = at_tid() & 63, tr = 4 * (lane >> 4), tc = lane & 15;
    for (int n0 = 0; n0 < N; n0 += 16) {
        if (!jobs.mine()) continue;
        const f32x4v acc = tile16(zero4(), G + n0, 1, sgm, min(16, N - n0), one, 0, 0, 16, M);
        if (tc == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (n0 + tr + i < N) part[n0 + tr + i] += acc[i];
        }
    }
}
__device__ __forceinline__ float wave_max(float v) {
    for (int o = 32; o; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
// softmax of a row, in place, by one wavefront; the row (n <= 256) stays in registers between the two reductions
__device__ __forceinline__ void softmax_row(float* a, int n) {
    const int lane = at_tid() & 63;
    float v[4], m = -INFINITY, t = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        v[u] = lane + 64 * u < n ? a[lane + 64 * u] : -INFINITY;
        m = fmaxf(m, v[u]);
    }
    m = wave_max(m);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        v[u] = lane + 64 * u < n ? expf(v[u] - m) : 0.f;
        t += v[u];
    }
    t = 1.0f / wave_sum_dpp(t);
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (lane + 64 * u < n) a[lane + 64 * u] = v[u] * t;
}
// d logit = a (d a - sum a d a), in place of d a
__device__ __forceinline__ void softmax_row_bwd(const float* a, float* da, int n) {
    const int lane = at_tid() & 63;
    float av[4], dv[4], c = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const bool in = lane + 64 * u < n;
        av[u] = in ? a[lane + 64 * u] : 0.f;
        dv[u] = in ? da[lane + 64 * u] : 0.f;
        c = fmaf(av[u], dv[u], c);
    }
    c = wave_sum_dpp(c);
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (lane + 64 * u < n) da[lane + 64 * u] = av[u] * (dv[u] - c);
}
// the P x P attention map: a row per lane, serially (10 elements; a wavefront per row would cost a row of the S-sized map each)
__device__ __forceinline__ void softmax_small(float* a, int n) {
    float m = -INFINITY, t = 0.f;
    for (int j = 0; j < n; ++j) m = fmaxf(m, a[j]);
    for (int j = 0; j < n; ++j) {
        const float e = expf(a[j] - m);
        a[j] = e;
        t += e;
    }
    t = 1.0f / t;
    for (int j = 0; j < n; ++j) a[j] *= t;
}
__device__ __forceinline__ void softmax_small_bwd(const float* a, float* da, int n) {
    float c = 0.f;
    for (int j = 0; j < n; ++j) c = fmaf(a[j], da[j], c);
    for (int j = 0; j < n; ++j) da[j] = a[j] * (da[j] - c);
}

// Register-row products of the per-sample phases: a lane holds a sample's row in registers, the weights are LDS broadcasts.  The loads of a
// batch of weight rows are issued together IN FRONT of their multiply-adds: hipcc otherwise emits load, wait, four multiply-adds, load, ...
// (measured: 40 k cycles for the MAM.linear step instead of 6 k -- every ds_read's latency exposed, with one wavefront per SIMD).
// o[j] += sum_k W[j K + k] x[k], j < NO
template <int K, int NO, int NJ>
__device__ __forceinline__ void rows_dot(const float* W, const float (&x)[K], float (&o)[NO]) {
#pragma unroll
    for (int j0 = 0; j0 < NO; j0 += NJ) {
        float4 w[NJ][K / 4];
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
            for (int k = 0; k < K / 4; ++k) w[jj][k] = reinterpret_cast<const float4*>(W + (j0 + jj) * K)[k];
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int k = 0; k < K / 4; ++k) {
                a0 = fmaf(w[jj][k].x, x[4 * k], a0); a1 = fmaf(w[jj][k].y, x[4 * k + 1], a1);
                a0 = fmaf(w[jj][k].z, x[4 * k + 2], a0); a1 = fmaf(w[jj][k].w, x[4 * k + 3], a1);
            }
            o[j0 + jj] += a0 + a1;
        }
    }
}
// o[n] += sum_m g[m] W[m ldw + n], n < NO (the transposed product in outer-product form: NM rows of W at a time)
template <int M, int NO, int NM>
__device__ __forceinline__ void cols_acc(const float* W, int ldw, const float (&g)[M], float (&o)[NO]) {
#pragma unroll
    for (int m0 = 0; m0 < M; m0 += NM) {
        float4 w[NM][NO / 4];
#pragma unroll
        for (int mm_ = 0; mm_ < NM; ++mm_)
#pragma unroll
            for (int n = 0; n < NO / 4; ++n) w[mm_][n] = reinterpret_cast<const float4*>(W + (m0 + mm_) * ldw)[n];
#pragma unroll
        for (int mm_ = 0; mm_ < NM; ++mm_)
#pragma unroll
            for (int n = 0; n < NO / 4; ++n) {
                o[4 * n] = fmaf(g[m0 + mm_], w[mm_][n].x, o[4 * n]); o[4 * n + 1] = fmaf(g[m0 + mm_], w[mm_][n].y, o[4 * n + 1]);
                o[4 * n + 2] = fmaf(g[m0 + mm_], w[mm_][n].z, o[4 * n + 2]); o[4 * n + 3] = fmaf(g[m0 + mm_], w[mm_][n].w, o[4 * n + 3]);
            }
    }
}
// the same over the P sub-exposures (a run-time count <= 16): o[n] += sum_p col[p cs] rows[p ld + n], n < NO
template <int NO>
__device__ __forceinline__ void p_acc(const float* col, int cs, const float* rows, int ld, int P, float (&o)[NO]) {
    for (int p0 = 0; p0 < P; p0 += 4) {
        float g[4];
        float4 w[4][NO / 4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int pp = p0 + u < P ? p0 + u : P - 1;
            g[u] = p0 + u < P ? col[pp * cs] : 0.f;
#pragma unroll
            for (int n = 0; n < NO / 4; ++n) w[u][n] = reinterpret_cast<const float4*>(rows + pp * ld)[n];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int n = 0; n < NO / 4; ++n) {
                o[4 * n] = fmaf(g[u], w[u][n].x, o[4 * n]); o[4 * n + 1] = fmaf(g[u], w[u][n].y, o[4 * n + 1]);
                o[4 * n + 2] = fmaf(g[u], w[u][n].z, o[4 * n + 2]); o[4 * n + 3] = fmaf(g[u], w[u][n].w, o[4 * n + 3]);
            }
    }
}
// out[p so] = rows[p ld .. + 16] . x for p = p0, p0 + AT_LPS, ... (the logits / their gradients of a sample against the sub-exposures' rows)
__device__ __forceinline__ void p_dot16(float* out, int so, const float* rows, int ld, int p0, int P, const float (&x)[AT_MID]) {
    for (int pa = p0; pa < P; pa += 2 * AT_LPS) {
        const int pb = pa + AT_LPS < P ? pa + AT_LPS : pa;
        float4 wa[4], wb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            wa[k] = reinterpret_cast<const float4*>(rows + pa * ld)[k];
            wb[k] = reinterpret_cast<const float4*>(rows + pb * ld)[k];
        }
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a = fmaf(wa[k].x, x[4 * k], a); a = fmaf(wa[k].y, x[4 * k + 1], a); a = fmaf(wa[k].z, x[4 * k + 2], a); a = fmaf(wa[k].w, x[4 * k + 3], a);
            b = fmaf(wb[k].x, x[4 * k], b); b = fmaf(wb[k].y, x[4 * k + 1], b); b = fmaf(wb[k].z, x[4 * k + 2], b); b = fmaf(wb[k].w, x[4 * k + 3], b);
        }
        out[pa * so] = a;
        if (pa + AT_LPS < P) out[pb * so] = b;
    }
}

__device__ __forceinline__ void tail_stage_weights(const TailKParams& p, const TailLds& L) {
    const TailDims& d = p.d;
    auto copy = [&](float* dst, const float* src, int n) {           // n is a multiple of 4 here except for the odd-width layer 0
        if ((n & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
#pragma unroll 4
            for (int i = threadIdx.x; i < n / 4; i += AT_NT) reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
        } else {
#pragma unroll 4
            for (int i = threadIdx.x; i < n; i += AT_NT) dst[i] = src[i];
        }
    };
    for (int l = 0; l < d.n_mot; ++l) {
        copy(L.mw(l), p.w[2 * l], tail_param_size(d, 2 * l));
        copy(L.mb(l), p.w[2 * l + 1], AT_CM);
    }
    const float* const* w = p.w + 2 * d.n_mot;
    copy(L.lin_w, w[TW_LIN_W], AT_CM * AT_WS); copy(L.lin_b, w[TW_LIN_B], AT_CM);
    copy(L.conva, w[TW_CONVA], AT_MID * AT_CM); copy(L.convb, w[TW_CONVB], AT_MID * AT_CM); copy(L.convc, w[TW_CONVC], AT_MID * AT_CM);
    copy(L.convn, w[TW_CONVN], AT_MID * AT_MID); copy(L.convl, w[TW_CONVL], AT_MID * AT_MID); copy(L.convd, w[TW_CONVD], AT_CM * AT_CM);
}

// The forward of one ray into LDS (both the forward kernel and the backward, which recomputes it).  On return (after the final
// barrier): x0, xs[], hi, li, kP, nP, q, aP, ls, kI, nI, aS, f, yb (= convd f, the BatchNorm's input).
__device__ __forceinline__ void tail_forward_ray(const TailKParams& p, const TailLds& L, long r, AtClock& clk) {
    const TailDims& d = p.d;
    const int tid = at_tid(), P = d.P, S = d.S;
    TileJobs jobs{0};
    // awp.py:89-95, 104-105: [integrated features | view_feature | direction encoding of the first sub-exposure's ray]
    for (int i = tid; i < P * AT_WS; i += AT_NT) {
        const int pp = i >> 6, k = i & 63;
        L.x0[pp * d.IN0 + k] = p.h[r * P * AT_WS + i];
        L.hi[i] = p.h_inter[r * P * AT_WS + i];
    }
    for (int i = tid; i < P * d.VC; i += AT_NT) {
        const int pp = i / d.VC, j = i - pp * d.VC;
        float v;
        if (j < d.VF) v = p.vf[r * d.VF + j];
        else {
            const int e = j - d.VF, axis = e % 3, blk = e / 3;          // blk 0: the direction; 1 + 2 k: sin(2^k d); 2 + 2 k: cos(2^k d)
            const float* rd = p.rays_d + r * P * 3;
            const float dn = rd[axis] / sqrtf(rd[0] * rd[0] + rd[1] * rd[1] + rd[2] * rd[2]);
            if (blk == 0) v = dn;
            else {
                const float arg = dn * (float)(1 << ((blk - 1) >> 1));
                v = ((blk - 1) & 1) ? cosf(arg) : sinf(arg);
            }
        }
        L.x0[pp * d.IN0 + AT_WS + j] = v;
    }
    __syncthreads();
    clk.mark(1);
    // awp.py:107-109 (layer 0) and mam.py:72-74 applied to the per-sample part's inter sums
    lin<1>(jobs, L.xs(0), AT_CM, L.x0, d.IN0, L.mw(0), L.mb(0), P, AT_CM, d.IN0);
    lin<0>(jobs, L.li, AT_CM, L.hi, AT_WS, L.lin_w, L.lin_b, P, AT_CM, AT_WS);
    // ... and to the intra sums [S, 64] -> ls [S, 32]: 16 x 16 tiles over (samples, channels), a wavefront per tile; a lane's four k-steps
    // of a sample row / a weight row are one 16-byte load (k = 16 jj + 4 (l / 16) + component on both operands)
    {
        const int wave = tid >> 6, lane = tid & 63, mn = lane & 15, kq = lane >> 4;
        const float* Wg = p.w[2 * d.n_mot + TW_LIN_W];
        const int nst = (S + 15) >> 4;
        for (int tt = wave; tt < 2 * nst; tt += AT_NT / 64) {
            const int st = tt >> 1, ct = tt & 1;
            const int srow = min(16 * st + mn, S - 1);
            const float4* xa = reinterpret_cast<const float4*>(p.h_intra + (r * S + srow) * AT_WS) + kq;
            const float4* wbp = reinterpret_cast<const float4*>(Wg + (16 * ct + mn) * AT_WS) + kq;
            float4 a[4], b[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) { a[jj] = xa[4 * jj]; b[jj] = wbp[4 * jj]; }
            f32x4v acc = zero4();
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jj].x, b[jj].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jj].y, b[jj].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jj].z, b[jj].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jj].w, b[jj].w, acc, 0, 0, 0);
            }
            const float bias = L.lin_b[16 * ct + mn];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int so = 16 * st + 4 * kq + i;
                if (so < S) L.ls[so * AT_LS + 16 * ct + mn] = acc[i] + bias;
            }
        }
    }
    const int sl = tid / AT_LPS, hf = tid % AT_LPS;
    __syncthreads();
    clk.mark(2);
    for (int l = 1; l < d.n_mot; ++l) {
        lin<1>(jobs, L.xs(l), AT_CM, L.xs(l - 1), AT_CM, L.mw(l), L.mb(l), P, AT_CM, AT_CM);
        if (l == 1) lin<0>(jobs, L.kP, AT_MID, L.li, AT_CM, L.conva, nullptr, P, AT_MID, AT_CM);            // mam.py:38
        __syncthreads();
        clk.mark(3);
    }
    if (d.n_mot == 1) {
        lin<0>(jobs, L.kP, AT_MID, L.li, AT_CM, L.conva, nullptr, P, AT_MID, AT_CM);
        __syncthreads();
        clk.mark(3);
    }
    const float* xg = L.xs(d.n_mot - 1);
    lin<0>(jobs, L.q, AT_MID, xg, AT_CM, L.convc, nullptr, P, AT_MID, AT_CM);                               // mam.py:41
    lin<0>(jobs, L.nP, AT_MID, L.kP, AT_MID, L.convn, nullptr, P, AT_MID, AT_MID);                          // mam.py:46
    for (int s0 = 0; s0 < S; s0 += AT_NT / AT_LPS) {                                                       // mam.py:39: convb
        const int s = s0 + sl;
        if (s < S) {
            float x[AT_CM];
#pragma unroll
            for (int c = 0; c < AT_CM; ++c) x[c] = L.ls[s * AT_LS + c];
            float o[AT_O16] = {};
            rows_dot<AT_CM, AT_O16, AT_O16>(L.convb + hf * AT_O16 * AT_CM, x, o);
#pragma unroll
            for (int j = 0; j < AT_O16; ++j) L.kI[s * AT_LK + hf * AT_O16 + j] = o[j];
        }
    }
    __syncthreads();
    clk.mark(5);
    mm<0, false>(jobs, L.aP, P, L.q, AT_MID, 1, L.kP, AT_MID, 1, nullptr, P, P, AT_MID);                   // mam.py:42: logits over the sub-exposures
    for (int s0 = 0; s0 < S; s0 += AT_NT / AT_LPS) {                                                       // mam.py:47 (convl), :43 (logits over the samples)
        const int s = s0 + sl;
        if (s < S) {
            float x[AT_MID];
#pragma unroll
            for (int c = 0; c < AT_MID; ++c) x[c] = L.kI[s * AT_LK + c];
            float o[AT_O16] = {};
            rows_dot<AT_MID, AT_O16, AT_O16>(L.convl + hf * AT_O16 * AT_MID, x, o);
#pragma unroll
            for (int j = 0; j < AT_O16; ++j) L.nI[s * AT_LK + hf * AT_O16 + j] = o[j];
            p_dot16(L.aS + s, d.SA, L.q, AT_MID, hf, P, x);
        }
    }
    __syncthreads();
    clk.mark(6);
    for (int row = tid >> 6; row < P; row += AT_NT / 64) softmax_row(L.aS + row * d.SA, S);          // mam.py:42-43: the softmaxes
    if (tid >= AT_NT - 64 && tid - (AT_NT - 64) < P) softmax_small(L.aP + (tid - (AT_NT - 64)) * P, P);
    __syncthreads();
    clk.mark(7);
    mm<0, false>(jobs, L.f, AT_CM, L.aP, P, 1, L.nP, 1, AT_MID, nullptr, P, AT_MID, P);                     // mam.py:49
    if ((tid >> 6) == AT_NT / 64 - 1) {                                                               // mam.py:50 (:52: the concatenation)
        const f32x4v acc = tile16(zero4(), L.aS, d.SA, 1, P, L.nI, AT_LK, 1, AT_MID, S);
        const int lane = tid & 63;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (4 * (lane >> 4) + i < P) L.f[(4 * (lane >> 4) + i) * AT_CM + AT_MID + (lane & 15)] = acc[i];
    }
    __syncthreads();
    clk.mark(8);
    lin<0>(jobs, L.yb, AT_CM, L.f, AT_CM, L.convd, nullptr, P, AT_CM, AT_CM);                               // mam.py:53: convd[0]
    __syncthreads();
    clk.mark(9);
}

__global__ __launch_bounds__(AT_NT) void k_awp_tail_fwd(TailKParams p) {
    extern __shared__ float lds[];
    TailLds L;
    tail_lds_layout(lds, p.d, false, L);
    AtClock clk;
    clk.start();
    tail_stage_weights(p, L);
    __syncthreads();
    clk.mark(0);
    const int tid = threadIdx.x, P = p.d.P;
    double s1 = 0.0, s2 = 0.0;                                   // lanes 0..31: the sums of this workgroup's y, y^2 of channel tid
    for (long r = blockIdx.x; r < p.R; r += gridDim.x) {
        tail_forward_ray(p, L, r, clk);
        if (p.saved) {                                           // the ray's forward state for the backward: 55 KB instead of a second forward
            float4* dst = reinterpret_cast<float4*>(p.saved + r * L.ray_floats);
            const float4* src = reinterpret_cast<const float4*>(L.x0);
#pragma unroll 4
            for (int i = tid; i < L.ray_floats / 4; i += AT_NT) dst[i] = src[i];
        }
        const float* xg = L.xs(p.d.n_mot - 1);
        for (int i = tid; i < P * AT_CM; i += AT_NT) {
            p.y[r * P * AT_CM + i] = L.yb[i];
            p.xg[r * P * AT_CM + i] = xg[i];
        }
        if (tid < AT_CM) {
            float a = 0.f, b = 0.f;
            for (int pp = 0; pp < P; ++pp) {
                const float v = L.yb[pp * AT_CM + tid];
                a += v;
                b = fmaf(v, v, b);
            }
            s1 += (double)a;
            s2 += (double)b;
        }
        __syncthreads();
        clk.mark(10);
    }
    clk.flush(p.stamps);
    if (tid < AT_CM) {
        p.bn_part[(long)blockIdx.x * 2 * AT_CM + tid] = s1;
        p.bn_part[(long)blockIdx.x * 2 * AT_CM + AT_CM + tid] = s2;
    }
}

struct TailFinishParams {
    const double* bn_part;
    int nparts, P, training, nblk;
    long R;
    float eps, momentum;
    const float *bn_w, *bn_b, *wl_w, *wl_b, *y, *xg;
    float *run_mean, *run_var;
    long long* num_batches;
    float *stats, *out;                // stats [64]: mean, 1 / sqrt(var + eps)
    // backward
    const float* d_out;
    float *dz, *partB;                 // partB [nblk][stride]: d beta [32], d gamma [32], d w_linear.weight [P, 32], d w_linear.bias [P]
    int partB_stride;
};

constexpr int AT_FT = 256, AT_FR = AT_FT / 32;      // threads and rays per workgroup of the finish kernels (a ray per 32 lanes)

// mam.py:53 (BatchNorm, residual, leaky_relu) and awp.py:112-115 of one ray on 32 lanes (lane = channel; called by every lane of the
// workgroup, rays past the end compute on the last ray and write nothing); returns through LDS: sh_hm[32] the mean over P, sh_w[P] the
// sigmoids, tot their sum
template <bool KEEP>
__device__ __forceinline__ void finish_ray(const TailFinishParams& p, long r, int c, const float* stat, float* sh_hm, float* sh_w, float& tot,
                                           float (&zkeep)[AT_MAXP]) {
    const int P = p.P;
    const float mean = stat[c], rstd = stat[AT_CM + c], g = p.bn_w[c], b = p.bn_b[c];
    float acc = 0.f;
#pragma unroll
    for (int pp = 0; pp < AT_MAXP; ++pp)
        if (pp < P) {
            const long at = (r * P + pp) * AT_CM + c;
            const float z = p.xg[at] + ((p.y[at] - mean) * rstd * g + b);
            if (KEEP) zkeep[pp] = z;
            acc += z > 0.f ? z : 0.2f * z;
        }
    sh_hm[c] = acc / (float)P;
    __syncthreads();
    if (c < P) {
        float a = p.wl_b[c];
        for (int k = 0; k < AT_CM; ++k) a = fmaf(p.wl_w[c * AT_CM + k], sh_hm[k], a);
        sh_w[c] = 1.0f / (1.0f + expf(-a));
    }
    __syncthreads();
    tot = 0.f;
    for (int j = 0; j < P; ++j) tot += sh_w[j];
}

__global__ __launch_bounds__(AT_FT) void k_awp_tail_finish(TailFinishParams p) {
    __shared__ double red[4][2 * AT_CM];
    __shared__ float stat[2 * AT_CM], hm[AT_FR][AT_CM], sw[AT_FR][AT_MAXP];
    const int tid = threadIdx.x, col = tid & 63, grp = tid >> 6;
    if (p.training) {
        double acc[4] = {0.0, 0.0, 0.0, 0.0};                   // four rows in flight per thread (every workgroup folds all the partial rows)
        int i = grp;
        for (; i + 12 < p.nparts; i += 16)
#pragma unroll
            for (int u2 = 0; u2 < 4; ++u2) acc[u2] += p.bn_part[(long)(i + 4 * u2) * 2 * AT_CM + col];
        for (; i < p.nparts; i += 4) acc[0] += p.bn_part[(long)i * 2 * AT_CM + col];
        red[grp][col] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        __syncthreads();
        if (tid < AT_CM) {
            const double n = (double)p.R * p.P;
            const double s1 = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
            const double s2 = red[0][AT_CM + tid] + red[1][AT_CM + tid] + red[2][AT_CM + tid] + red[3][AT_CM + tid];
            const double mu = s1 / n;
            double var = s2 / n - mu * mu;
            var = var > 0.0 ? var : 0.0;
            stat[tid] = (float)mu;
            stat[AT_CM + tid] = 1.0f / sqrtf((float)var + p.eps);
            if (blockIdx.x == 0) {
                if (p.run_mean) {                                           // BatchNorm1d's running estimates (momentum blend, unbiased variance)
                    p.run_mean[tid] = (1.0f - p.momentum) * p.run_mean[tid] + p.momentum * (float)mu;
                    p.run_var[tid] = (1.0f - p.momentum) * p.run_var[tid] + p.momentum * (float)(var * (n > 1.0 ? n / (n - 1.0) : 1.0));
                }
                if (tid == 0 && p.num_batches) *p.num_batches += 1;
            }
        }
    } else if (tid < AT_CM) {
        stat[tid] = p.run_mean[tid];
        stat[AT_CM + tid] = 1.0f / sqrtf(p.run_var[tid] + p.eps);
    }
    __syncthreads();
    if (blockIdx.x == 0 && tid < 2 * AT_CM) p.stats[tid] = stat[tid];
    const int slot = tid >> 5, c = tid & 31;
    const long r = (long)blockIdx.x * AT_FR + slot;
    float tot, unused[AT_MAXP];
    finish_ray<false>(p, r < p.R ? r : p.R - 1, c, stat, hm[slot], sw[slot], tot, unused);
    if (r < p.R && c < p.P) p.out[r * p.P + c] = sw[slot][c] / tot;
}

// d out -> d z, and this workgroup's partial sums of d beta, d gamma (BatchNorm) and d w_linear
__global__ __launch_bounds__(AT_FT) void k_awp_tail_bwd0(TailFinishParams p) {
    __shared__ float stat[2 * AT_CM], hm[AT_FR][AT_CM], sw[AT_FR][AT_MAXP], dpre[AT_FR][AT_MAXP], red[AT_FR][2 * AT_CM];
    const int tid = threadIdx.x, P = p.P;
    if (tid < 2 * AT_CM) stat[tid] = p.stats[tid];
    __syncthreads();
    const int slot = tid >> 5, c = tid & 31;
    const long r = (long)blockIdx.x * AT_FR + slot;
    float dbeta = 0.f, dgamma = 0.f;
    for (int j = tid; j < AT_FR * AT_MAXP; j += AT_FT) (&dpre[0][0])[j] = 0.f;
    float tot, z[AT_MAXP];
    finish_ray<true>(p, r < p.R ? r : p.R - 1, c, stat, hm[slot], sw[slot], tot, z);
    if (r >= p.R) hm[slot][c] = 0.f;                           // (rays past the end add nothing to d w_linear)
    if (r < p.R) {
        // out_j = w_j / tot:  d w_j = (g_j - sum_k g_k out_k) / tot;  d pre_j = d w_j w_j (1 - w_j)
        float dot = 0.f;
        for (int j = 0; j < P; ++j) dot = fmaf(p.d_out[r * P + j], sw[slot][j] / tot, dot);
        if (c < P) {
            const float w = sw[slot][c];
            dpre[slot][c] = (p.d_out[r * P + c] - dot) / tot * w * (1.0f - w);
        }
    }
    __syncthreads();
    if (r < p.R) {
        float dhm = 0.f;
        for (int j = 0; j < P; ++j) dhm = fmaf(dpre[slot][j], p.wl_w[j * AT_CM + c], dhm);
        dhm /= (float)P;
        const float mean = stat[c], rstd = stat[AT_CM + c];
#pragma unroll
        for (int pp = 0; pp < AT_MAXP; ++pp)
            if (pp < P) {
                const long at = (r * P + pp) * AT_CM + c;
                const float dzv = z[pp] > 0.f ? dhm : 0.2f * dhm;
                p.dz[at] = dzv;
                dbeta += dzv;
                dgamma = fmaf(dzv, (p.y[at] - mean) * rstd, dgamma);
            }
    }
    red[slot][c] = dbeta;
    red[slot][AT_CM + c] = dgamma;
    __syncthreads();
    float* part = p.partB + (long)blockIdx.x * p.partB_stride;
    if (tid < 2 * AT_CM) {
        float a = 0.f;
        for (int s = 0; s < AT_FR; ++s) a += red[s][tid];
        part[tid] = a;
    }
    for (int i = tid; i < P * AT_CM + P; i += AT_FT) {          // d w_linear.weight[j][k] = sum_rays d pre_j hm_k;  .bias[j] = sum d pre_j
        float a = 0.f;
        if (i < P * AT_CM) {
            const int j = i / AT_CM, k = i - j * AT_CM;
            for (int s = 0; s < AT_FR; ++s) a = fmaf(dpre[s][j], hm[s][k], a);
        } else {
            for (int s = 0; s < AT_FR; ++s) a += dpre[s][i - P * AT_CM];
        }
        part[2 * AT_CM + i] = a;
    }
}

__global__ __launch_bounds__(AT_NT) void k_awp_tail_bwd(TailKParams p) {
    extern __shared__ float lds[];
    __shared__ double redd[AT_NT / 64][2 * AT_CM];
    __shared__ float sums[2 * AT_CM], stat[2 * AT_CM];
    TailLds L;
    tail_lds_layout(lds, p.d, true, L);
    AtClock clk;
    clk.start();
    tail_stage_weights(p, L);
    const TailDims& d = p.d;
    const int tid0 = threadIdx.x, P = d.P, S = d.S, nm = d.n_mot;
    {   // d beta, d gamma over all rays (k_awp_tail_bwd0's partial rows)
        const int col = tid0 & 63, grp = tid0 >> 6;
        double a = 0.0;
        for (int i = grp; i < p.nblkB; i += AT_NT / 64) a += (double)p.partB[(long)i * p.partB_stride + col];
        redd[grp][col] = a;
        if (tid0 < 2 * AT_CM) stat[tid0] = p.stats[tid0];
    }
    float* part = p.partA + (long)blockIdx.x * p.partA_stride;
    if (tid0 == 0) L.tmp[63] = 1.0f;                        // the column of ones of the bias-gradient tiles
    for (long i = tid0; i < p.off[p.nw]; i += AT_NT) part[i] = 0.f;
    __syncthreads();
    if (tid0 < 2 * AT_CM) {
        double a = 0.0;
        for (int g = 0; g < AT_NT / 64; ++g) a += redd[g][tid0];
        sums[tid0] = (float)(a / ((double)p.R * P));
    }
    __syncthreads();
    const long* off = p.off;
    const int wb = 2 * nm;
    for (long r = blockIdx.x; r < p.R; r += gridDim.x) {
        const int tid = at_tid(), sl = tid / AT_LPS, hf = tid % AT_LPS;      // (per ray: see at_tid)
        TileJobs jobs{0};
        if (p.saved) {
            const float4* src = reinterpret_cast<const float4*>(p.saved + r * L.ray_floats);
            float4* dst = reinterpret_cast<float4*>(L.x0);
            const int n4 = L.ray_floats / 4;
            for (int i0 = 0; i0 < n4; i0 += 8 * AT_NT) {         // eight 16-byte loads per thread in flight (the block is one trip at P = 10, S = 128)
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + tid + u * AT_NT;
                    v[u] = src[i < n4 ? i : 0];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + tid + u * AT_NT;
                    if (i < n4) dst[i] = v[u];
                }
            }
            __syncthreads();
            clk.mark(1);
        } else {
            tail_forward_ray(p, L, r, clk);
        }
        const float* xg = L.xs(nm - 1);
        // BatchNorm backward (mam.py:24-27 in training: batch statistics; eval: the running estimates are constants), and the residual
        for (int i = tid; i < P * AT_CM; i += AT_NT) {
            const int c = i & 31;
            const float dzv = p.dz[r * P * AT_CM + i];
            const float g = p.w[wb + TW_BN_W][c] * stat[AT_CM + c];
            const float yh = (L.yb[i] - stat[c]) * stat[AT_CM + c];
            L.yb[i] = p.training ? g * (dzv - sums[c] - yh * sums[AT_CM + c]) : g * dzv;          // d y
            L.dxa[i] = dzv;                                                                         // d x_global, the residual's share
        }
        __syncthreads();
        clk.mark(11);
        wacc(jobs, part + off[wb + TW_CONVD], L.yb, AT_CM, L.f, AT_CM, P, AT_CM, AT_CM);
        mm<0, false>(jobs, L.df, AT_CM, L.yb, AT_CM, 1, L.convd, 1, AT_CM, nullptr, P, AT_CM, AT_CM);      // d f = d y convd
        __syncthreads();
        clk.mark(12);
        // attention over the sub-exposures (mam.py:42, 46, 49) and over the samples (:43, 47, 50)
        mm<0, false>(jobs, L.daP, P, L.df, AT_CM, 1, L.nP, AT_MID, 1, nullptr, P, P, AT_MID);              // d aP[p][p'] = d fP[p] . nP[p']
        mm<0, false>(jobs, L.dnP, AT_MID, L.aP, 1, P, L.df, 1, AT_CM, nullptr, P, AT_MID, P);              // d nP[p'][m] = sum_p aP[p][p'] d fP[p][m]
        for (int s0 = 0; s0 < S; s0 += AT_NT / AT_LPS) {
            const int s = s0 + sl;
            if (s < S) {
                float x[AT_MID];
#pragma unroll
                for (int c = 0; c < AT_MID; ++c) x[c] = L.nI[s * AT_LK + c];
                p_dot16(L.daS + s, d.SA, L.df + AT_MID, AT_CM, hf, P, x);                            // d aS[p][s] = d fI[p] . nI[s]
            }
        }
        __syncthreads();
        clk.mark(13);
        for (int s0 = 0; s0 < S; s0 += AT_NT / AT_LPS) {                                                  // d nI[s][m] = sum_p aS[p][s] d fI[p][m]  (into nI's place)
            const int s = s0 + sl;
            if (s < S) {
                float acc[AT_O16] = {};
                p_acc<AT_O16>(L.aS + s, d.SA, L.df + AT_MID + hf * AT_O16, AT_CM, P, acc);
#pragma unroll
                for (int j = 0; j < AT_O16; ++j) L.nI[s * AT_LK + hf * AT_O16 + j] = acc[j];
            }
        }
        for (int row = tid >> 6; row < P; row += AT_NT / 64) softmax_row_bwd(L.aS + row * d.SA, L.daS + row * d.SA, S);
        if (tid >= AT_NT - 64 && tid - (AT_NT - 64) < P) softmax_small_bwd(L.aP + (tid - (AT_NT - 64)) * P, L.daP + (tid - (AT_NT - 64)) * P, P);
        __syncthreads();
        clk.mark(14);
        float* dnI = L.nI;
        // d q = d lgP kP + d lgS kI;  d kP = d lgP^T q + d nP convn;  d kI = d lgS^T q + d nI convl
        const int wave = tid >> 6, lane = tid & 63, tr = 4 * (lane >> 4), tc = lane & 15;       // a tile16 result: rows tr .. tr + 3, column tc
        if (wave == 0) {
            f32x4v acc = tile16(zero4(), L.daS, d.SA, 1, P, L.kI, AT_LK, 1, AT_MID, S);
            acc = tile16(acc, L.daP, P, 1, P, L.kP, AT_MID, 1, AT_MID, P);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (tr + i < P) L.dq[(tr + i) * AT_MID + tc] = acc[i];
        }
        if (jobs.mine()) {                                                                            // d kP = d lgP^T q + d nP convn (one tile, one owner)
            f32x4v acc = tile16(zero4(), L.daP, 1, P, P, L.q, AT_MID, 1, AT_MID, P);
            acc = tile16(acc, L.dnP, AT_MID, 1, P, L.convn, AT_MID, 1, AT_MID, AT_MID);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (tr + i < P) L.dkP[(tr + i) * AT_MID + tc] = acc[i];
        }
        wacc(jobs, part + off[wb + TW_CONVN], L.dnP, AT_MID, L.kP, AT_MID, P, AT_MID, AT_MID);
        if (wave == 1) {                                                                              // d convl[m'][m] = sum_s d nI[s][m'] kI[s][m]
            const f32x4v acc = tile16(zero4(), dnI, 1, AT_LK, AT_MID, L.kI, AT_LK, 1, AT_MID, S);
#pragma unroll
            for (int i = 0; i < 4; ++i) part[off[wb + TW_CONVL] + (tr + i) * AT_MID + tc] += acc[i];
        }
        for (int s0 = 0; s0 < S; s0 += AT_NT / AT_LPS) {
            const int s = s0 + sl;
            if (s < S) {
                float acc[AT_O16] = {}, g[AT_MID];
                p_acc<AT_O16>(L.daS + s, d.SA, L.q + hf * AT_O16, AT_MID, P, acc);
#pragma unroll
                for (int m = 0; m < AT_MID; ++m) g[m] = dnI[s * AT_LK + m];
                cols_acc<AT_MID, AT_O16, 8>(L.convl + hf * AT_O16, AT_MID, g, acc);
#pragma unroll
                for (int j = 0; j < AT_O16; ++j) L.dkI[s * AT_LK + hf * AT_O16 + j] = acc[j];
            }
        }
        __syncthreads();
        clk.mark(15);
        // conva / convb / convc, and back through MAM.linear
        wacc(jobs, part + off[wb + TW_CONVA], L.dkP, AT_MID, L.li, AT_CM, P, AT_MID, AT_CM);
        if (wave == 2 || wave == 3) {                                                                 // d convb[m][c] = sum_s d kI[s][m] ls[s][c]
            const int ct = wave - 2;
            const f32x4v acc = tile16(zero4(), L.dkI, 1, AT_LK, AT_MID, L.ls + 16 * ct, AT_LS, 1, 16, S);
#pragma unroll
            for (int i = 0; i < 4; ++i) part[off[wb + TW_CONVB] + (tr + i) * AT_CM + 16 * ct + tc] += acc[i];
        }
        wacc(jobs, part + off[wb + TW_CONVC], L.dq, AT_MID, xg, AT_CM, P, AT_MID, AT_CM);
        mm<0, false>(jobs, L.dli, AT_CM, L.dkP, AT_MID, 1, L.conva, 1, AT_CM, nullptr, P, AT_CM, AT_MID);
        mm<0, true>(jobs, L.dxa, AT_CM, L.dq, AT_MID, 1, L.convc, 1, AT_CM, nullptr, P, AT_CM, AT_MID);   // d x_global += d q convc
        for (int s0 = 0; s0 < S; s0 += AT_NT / AT_LPS) {                                                  // d ls[s] = d kI[s] convb
            const int s = s0 + sl;
            if (s < S) {
                float acc[AT_O32] = {}, g[AT_MID];
#pragma unroll
                for (int m = 0; m < AT_MID; ++m) g[m] = L.dkI[s * AT_LK + m];
                cols_acc<AT_MID, AT_O32, 8>(L.convb + hf * AT_O32, AT_CM, g, acc);
#pragma unroll
                for (int j = 0; j < AT_O32; ++j) L.dls[s * AT_LS + hf * AT_O32 + j] = acc[j];
            }
        }
        __syncthreads();
        clk.mark(16);
        {   // d MAM.linear.weight [32][64] += d ls^T h_intra + d li^T h_inter: eight 16 x 16 tiles, one per wavefront (AT_NT / 64 == 8)
            const int ct = wave & 1, kt = wave >> 1;
            float* dst = part + off[wb + TW_LIN_W] + (16 * ct + tr) * AT_WS + 16 * kt + tc;
            float old[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) old[i] = dst[i * AT_WS];
            f32x4v acc = zero4();
            const float* xg_ = p.h_intra + r * S * AT_WS + 16 * kt + tc;          // B[k = sample][n]: 64 contiguous bytes per sample and tile
            for (int s0 = 0; s0 < S; s0 += 128) {                                 // 32 samples-of-four: every global operand requested first
                float bv[32];
#pragma unroll
                for (int u = 0; u < 32; ++u) {
                    const int sx = s0 + 4 * u + (lane >> 4);
                    bv[u] = xg_[(long)(sx < S ? sx : S - 1) * AT_WS];
                }
#pragma unroll
                for (int u = 0; u < 32; ++u) {
                    const int sx = s0 + 4 * u + (lane >> 4);
                    const float av = sx < S ? L.dls[sx * AT_LS + 16 * ct + tc] : 0.f;
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, sx < S ? bv[u] : 0.f, acc, 0, 0, 0);
                }
            }
            acc = tile16(acc, L.dli + 16 * ct, 1, AT_CM, 16, L.hi + 16 * kt, AT_WS, 1, 16, P);
#pragma unroll
            for (int i = 0; i < 4; ++i) dst[i * AT_WS] = old[i] + acc[i];
        }
        for (int n0 = 0; n0 < AT_CM; n0 += 16)                                                        // d MAM.linear.bias = the column sums of d li and d ls
            if (jobs.mine()) {
                f32x4v acc = tile16(zero4(), L.dli + n0, 1, AT_CM, 16, L.tmp + 63, 0, 0, 16, P);
                acc = tile16(acc, L.dls + n0, 1, AT_LS, 16, L.tmp + 63, 0, 0, 16, S);
                if (tc == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) part[off[wb + TW_LIN_B] + n0 + tr + i] += acc[i];
                }
            }
        mm<0, false>(jobs, p.d_hi + r * P * AT_WS, AT_WS, L.dli, AT_CM, 1, L.lin_w, 1, AT_WS, nullptr, P, AT_WS, AT_CM);
        {   // d h_intra [S][64] = d ls MAM.linear.weight: tiles over (samples, k), K = 32
            const int nst = (S + 15) >> 4;
            for (int tt = wave; tt < 4 * nst; tt += AT_NT / 64) {
                const int st = tt >> 2, kt = tt & 3;
                const f32x4v acc = tile16(zero4(), L.dls + 16 * st * AT_LS, AT_LS, 1, min(16, S - 16 * st), L.lin_w + 16 * kt, AT_WS, 1, 16, AT_CM);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int so = 16 * st + tr + i;
                    if (so < S) p.d_hs[(r * S + so) * AT_WS + 16 * kt + tc] = acc[i];
                }
            }
        }
        // the motion embedding backwards (awp.py:107-109)
        float *dx = L.dxa, *dprev = L.dxb;
        for (int l = nm - 1; l >= 0; --l) {
            __syncthreads();
            clk.mark(l == nm - 1 ? 17 : 19);
            for (int i = tid; i < P * AT_CM; i += AT_NT) dx[i] = L.xs(l)[i] > 0.f ? dx[i] : 0.f;
            __syncthreads();
            clk.mark(18);
            const int K = l == 0 ? d.IN0 : AT_CM;
            const float* xin = l == 0 ? L.x0 : L.xs(l - 1);
            wacc(jobs, part + off[2 * l], dx, AT_CM, xin, K, P, AT_CM, K);
            bacc(jobs, part + off[2 * l + 1], dx, AT_CM, P, AT_CM, L.tmp + 63);
            mm<0, false>(jobs, dprev, K, dx, AT_CM, 1, L.mw(l), 1, K, nullptr, P, K, AT_CM);
            float* t = dx; dx = dprev; dprev = t;
        }
        __syncthreads();
        clk.mark(19);
        // dx: d [integrated features | view_feature | direction encoding] of every sub-exposure, [P][IN0]
        for (int i = tid; i < P * AT_WS; i += AT_NT) p.d_h[r * P * AT_WS + i] = dx[(i >> 6) * d.IN0 + (i & 63)];
        for (int j = tid; j < d.VC; j += AT_NT) {
            float a = 0.f;
            for (int pp = 0; pp < P; ++pp) a += dx[pp * d.IN0 + AT_WS + j];
            if (j < d.VF) p.d_vf[r * d.VF + j] = a;
            else L.tmp[j - d.VF] = a;
        }
        __syncthreads();
        clk.mark(20);
        if (tid < 3 * P) {
            float out = 0.f;
            if (tid < 3 && d.F >= 0) {                                 // awp.py:89-92: d (d / |d|) of the first sub-exposure's ray
                const float* rd = p.rays_d + r * P * 3;
                const float nrm = sqrtf(rd[0] * rd[0] + rd[1] * rd[1] + rd[2] * rd[2]);
                float mine = 0.f, dotv = 0.f;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const float dn = rd[a] / nrm;
                    float g = L.tmp[a];
                    for (int k = 0; k < d.F; ++k) {
                        const float fr = (float)(1 << k), arg = dn * fr;
                        g = fmaf(L.tmp[3 + 6 * k + a], fr * cosf(arg), g);
                        g = fmaf(L.tmp[6 + 6 * k + a], -fr * sinf(arg), g);
                    }
                    mine = a == tid ? g : mine;
                    dotv = fmaf(g, dn, dotv);
                }
                out = (mine - rd[tid] / nrm * dotv) / nrm;
            }
            p.d_rays[r * P * 3 + tid] = out;
        }
        __syncthreads();
        clk.mark(21);
    }
    clk.flush(p.stamps);
}

struct TailReduceParams {
    const float *partA, *partB;
    int nA, nB, partB_stride, n_tail, P;          // n_tail: elements of d_params in front of convd.1.weight
    long total, partA_stride;
    float* d_params;
};
// d_params[i] = sum over the partial rows; the BatchNorm and w_linear gradients come from k_awp_tail_bwd0's rows (d beta, d gamma, d W, d b).
// 64 parameters per workgroup, its four wavefronts each a quarter of the rows (four rows in flight per thread), folded through LDS.
__global__ __launch_bounds__(256) void k_awp_tail_reduce(TailReduceParams p) {
    __shared__ float fold[4][64];
    const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + col;
    float a = 0.f;
    if (i < p.total) {
        const bool tail = i < p.n_tail;
        const long j = i - p.n_tail;                                   // 0..31 d gamma, 32..63 d beta, then w_linear
        const long c = tail ? i : (j < AT_CM ? AT_CM + j : (j < 2 * AT_CM ? j - AT_CM : j));
        const float* src = tail ? p.partA : p.partB;
        const long stride = tail ? p.partA_stride : p.partB_stride;
        const int n = tail ? p.nA : p.nB;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        int b = grp;
        for (; b + 12 < n; b += 16)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] += src[(long)(b + 4 * u) * stride + c];
        for (; b < n; b += 4) acc[0] += src[(long)b * stride + c];
        a = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    }
    fold[grp][col] = a;
    __syncthreads();
    if (grp == 0 && i < p.total) p.d_params[i] = (fold[0][col] + fold[1][col]) + (fold[2][col] + fold[3][col]);
}

}  // namespace evd

using namespace evd;

static int tail_check(const char* who, const evd_awp_tail_desc* d, long R, bool bwd, TailDims& dims, size_t& lds_bytes) {
    EVD_REQUIRE(d, "%s: null descriptor", who);
    EVD_REQUIRE(R >= 0 && d->P >= 1 && d->P <= AT_MAXP && d->S >= 1, "%s: P = %d (built: 1..%d), S = %d", who, d->P, AT_MAXP, d->S);
    EVD_REQUIRE(d->n_mot >= 1 && d->n_mot <= AT_MAXMOT, "%s: %d motion embedding layers (built: 1..%d)", who, d->n_mot, AT_MAXMOT);
    EVD_REQUIRE(d->VF >= 0 && d->VF <= 64 && d->dir_freqs >= -1 && d->dir_freqs <= 4, "%s: view_feature width %d (<= 64), dir_freqs %d (-1..4)",
                who, d->VF, d->dir_freqs);
    dims = tail_dims(d->P, d->S, d->VF, d->dir_freqs, d->n_mot);
    TailLds L;
    lds_bytes = sizeof(float) * tail_lds_layout(nullptr, dims, bwd, L);
    const size_t fixed = bwd ? (AT_NT / 64) * 64 * sizeof(double) + 1024 : 0;       // the backward kernel's static arrays
    EVD_REQUIRE(lds_bytes + fixed <= 160 * 1024, "%s: P = %d, S = %d needs %zu bytes of LDS per ray (the CU has 160 KiB)", who, d->P, d->S,
                lds_bytes + fixed);
    return EVD_OK;
}

static int tail_grid(long R) {
    int cus = 256;
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    return (int)std::min<long>(R, cus);
}

static void tail_fill(TailKParams& k, const TailDims& dims, const evd_awp_tail_desc* d, const float* const* params, long R) {
    k.d = dims;
    k.nw = 2 * dims.n_mot + TW_COUNT;
    long at = 0;
    for (int i = 0; i < k.nw; ++i) {
        k.w[i] = params[i];
        k.off[i] = at;
        at += tail_param_size(dims, i);
    }
    k.off[k.nw] = at;
    k.training = d->training;
    k.eps = d->bn_eps;
    k.momentum = d->bn_momentum;
    k.R = R;
}

static int tail_nblk(long R) { return (int)((R + AT_FR - 1) / AT_FR); }
static int tail_partB_stride(int P) { return 2 * AT_CM + P * AT_CM + P; }

extern "C" {

int evd_awp_tail_num_params(int n_mot) { return 2 * n_mot + TW_COUNT; }

long evd_awp_tail_param_count(const evd_awp_tail_desc* d) {
    if (!d || d->n_mot < 1 || d->n_mot > AT_MAXMOT) return -1;
    const TailDims dims = tail_dims(d->P, d->S, d->VF, d->dir_freqs, d->n_mot);
    long at = 0;
    for (int i = 0; i < 2 * d->n_mot + TW_COUNT; ++i) at += tail_param_size(dims, i);
    return at;
}

long evd_awp_tail_saved_floats(const evd_awp_tail_desc* d) {
    if (!d || d->n_mot < 1 || d->n_mot > AT_MAXMOT || d->P < 1 || d->P > AT_MAXP || d->S < 1) return -1;
    TailLds L;
    tail_lds_layout(nullptr, tail_dims(d->P, d->S, d->VF, d->dir_freqs, d->n_mot), false, L);
    return L.ray_floats;
}

size_t evd_awp_tail_workspace_bytes(const evd_awp_tail_desc* d, long R, int backward) {
    if (!d || R < 0) return 0;
    const int grid = tail_grid(R > 0 ? R : 1);
    if (!backward) return sizeof(double) * 2 * AT_CM * (size_t)grid + 1024;
    const long total = evd_awp_tail_param_count(d);
    return sizeof(float) * ((size_t)R * d->P * AT_CM + (size_t)tail_nblk(R) * tail_partB_stride(d->P) + (size_t)grid * (size_t)((total + 3) & ~3L)) + 2048;
}

int evd_awp_tail_forward(const evd_awp_tail_desc* d, const float* const* params, const float* h, const float* view_feature,
                         const float* rays_d, const float* h_inter, const float* h_intra, long R, float* bn_running_mean,
                         float* bn_running_var, long long* bn_num_batches, float* out, float* saved_y, float* saved_xg, float* saved_stats,
                         float* saved_rays, void* workspace, size_t workspace_bytes, void* stream) {
    TailDims dims;
    size_t lds = 0;
    if (d && d->training) {          // a training-mode forward is followed by the backward: refuse here what that kernel's working set cannot hold
        if (int e = tail_check("evd_awp_tail_forward", d, R, true, dims, lds)) return e;
    }
    if (int e = tail_check("evd_awp_tail_forward", d, R, false, dims, lds)) return e;
    EVD_REQUIRE(params && h && rays_d && h_inter && h_intra && out && saved_y && saved_xg && saved_stats && workspace,
                "evd_awp_tail_forward: null argument");
    EVD_REQUIRE(dims.VF == 0 || view_feature, "evd_awp_tail_forward: view_feature missing (VF = %d)", dims.VF);
    EVD_REQUIRE(d->training || (bn_running_mean && bn_running_var), "evd_awp_tail_forward: eval mode needs the BatchNorm running estimates");
    EVD_REQUIRE(!bn_running_mean == !bn_running_var, "evd_awp_tail_forward: bn_running_mean and bn_running_var go together");
    for (int i = 0; i < evd_awp_tail_num_params(dims.n_mot); ++i) EVD_REQUIRE(params[i], "evd_awp_tail_forward: parameter %d missing", i);
    EVD_REQUIRE(workspace_bytes >= evd_awp_tail_workspace_bytes(d, R, 0), "evd_awp_tail_forward: workspace too small");
    if (R == 0) return EVD_OK;
    TailKParams k{};
    tail_fill(k, dims, d, params, R);
    k.h = h; k.vf = view_feature; k.rays_d = rays_d; k.h_inter = h_inter; k.h_intra = h_intra;
    k.y = saved_y; k.xg = saved_xg; k.saved = saved_rays;
    k.bn_part = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(workspace) + 63) & ~(uintptr_t)63);
    k.stamps = reinterpret_cast<long long*>(static_cast<char*>(workspace) + evd_awp_tail_workspace_bytes(d, R, 0) - 512);
    const int grid = tail_grid(R);
    EVD_SET_MAX_LDS(k_awp_tail_fwd, 160 * 1024);
    k_awp_tail_fwd<<<grid, AT_NT, lds, as_stream(stream)>>>(k);
    EVD_HIP(hipGetLastError());
    TailFinishParams f{};
    f.bn_part = k.bn_part; f.nparts = grid; f.P = dims.P; f.training = d->training; f.R = R; f.eps = d->bn_eps; f.momentum = d->bn_momentum;
    const float* const* w = params + 2 * dims.n_mot;
    f.bn_w = w[TW_BN_W]; f.bn_b = w[TW_BN_B]; f.wl_w = w[TW_WL_W]; f.wl_b = w[TW_WL_B];
    f.y = saved_y; f.xg = saved_xg; f.run_mean = bn_running_mean; f.run_var = bn_running_var; f.num_batches = bn_num_batches;
    f.stats = saved_stats; f.out = out;
    k_awp_tail_finish<<<tail_nblk(R), AT_FT, 0, as_stream(stream)>>>(f);
    EVD_HIP(hipGetLastError());
    return EVD_OK;
}

int evd_awp_tail_backward(const evd_awp_tail_desc* d, const float* const* params, const float* h, const float* view_feature,
                          const float* rays_d, const float* h_inter, const float* h_intra, long R, const float* saved_y,
                          const float* saved_xg, const float* saved_stats, const float* saved_rays, const float* d_out, float* d_h, float* d_view_feature,
                          float* d_rays_d, float* d_h_inter, float* d_h_intra, float* d_params, void* workspace, size_t workspace_bytes,
                          void* stream) {
    TailDims dims;
    size_t lds = 0;
    if (int e = tail_check("evd_awp_tail_backward", d, R, true, dims, lds)) return e;
    EVD_REQUIRE(params && h && rays_d && h_inter && h_intra && saved_y && saved_xg && saved_stats && d_out && d_h && d_rays_d && d_h_inter &&
                    d_h_intra && d_params && workspace,
                "evd_awp_tail_backward: null argument");
    EVD_REQUIRE(dims.VF == 0 || (view_feature && d_view_feature), "evd_awp_tail_backward: view_feature / d_view_feature missing (VF = %d)", dims.VF);
    for (int i = 0; i < evd_awp_tail_num_params(dims.n_mot); ++i) EVD_REQUIRE(params[i], "evd_awp_tail_backward: parameter %d missing", i);
    EVD_REQUIRE(workspace_bytes >= evd_awp_tail_workspace_bytes(d, R, 1), "evd_awp_tail_backward: workspace too small");
    const long total = evd_awp_tail_param_count(d);
    if (R == 0) {
        EVD_HIP(hipMemsetAsync(d_params, 0, sizeof(float) * (size_t)total, as_stream(stream)));
        return EVD_OK;
    }
    const int grid = tail_grid(R), nblk = tail_nblk(R), strideB = tail_partB_stride(dims.P);
    float* dz = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 63) & ~(uintptr_t)63);
    float* partB = dz + (size_t)R * dims.P * AT_CM;
    float* partA = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(partB + (size_t)nblk * strideB) + 63) & ~(uintptr_t)63);
    const float* const* w = params + 2 * dims.n_mot;
    TailFinishParams f{};
    f.P = dims.P; f.training = d->training; f.R = R; f.eps = d->bn_eps; f.nblk = nblk;
    f.bn_w = w[TW_BN_W]; f.bn_b = w[TW_BN_B]; f.wl_w = w[TW_WL_W]; f.wl_b = w[TW_WL_B];
    f.y = saved_y; f.xg = saved_xg; f.stats = const_cast<float*>(saved_stats); f.d_out = d_out; f.dz = dz; f.partB = partB; f.partB_stride = strideB;
    k_awp_tail_bwd0<<<nblk, AT_FT, 0, as_stream(stream)>>>(f);
    EVD_HIP(hipGetLastError());
    TailKParams k{};
    tail_fill(k, dims, d, params, R);
    k.h = h; k.vf = view_feature; k.rays_d = rays_d; k.h_inter = h_inter; k.h_intra = h_intra;
    k.y = const_cast<float*>(saved_y); k.xg = const_cast<float*>(saved_xg); k.saved = const_cast<float*>(saved_rays);
    k.dz = dz; k.stats = saved_stats; k.partB = partB; k.nblkB = nblk; k.partB_stride = strideB;
    k.partA_stride = (total + 3) & ~3L;
    k.stamps = reinterpret_cast<long long*>(static_cast<char*>(workspace) + evd_awp_tail_workspace_bytes(d, R, 1) - 512);
    k.d_h = d_h; k.d_vf = d_view_feature; k.d_rays = d_rays_d; k.d_hi = d_h_inter; k.d_hs = d_h_intra; k.partA = partA;
    EVD_SET_MAX_LDS(k_awp_tail_bwd, 160 * 1024 - (AT_NT / 64) * 64 * sizeof(double) - 1024);
    k_awp_tail_bwd<<<grid, AT_NT, lds, as_stream(stream)>>>(k);
    EVD_HIP(hipGetLastError());
    TailReduceParams rp{};
    rp.partA = partA; rp.partB = partB; rp.nA = grid; rp.nB = nblk; rp.partB_stride = strideB; rp.P = dims.P; rp.total = total;
    rp.partA_stride = k.partA_stride;
    rp.n_tail = (int)k.off[2 * dims.n_mot + TW_BN_W];
    rp.d_params = d_params;
    k_awp_tail_reduce<<<(unsigned)((total + 63) / 64), 256, 0, as_stream(stream)>>>(rp);
    EVD_HIP(hipGetLastError());
    return EVD_OK;
}

}  // extern "C"
