// PDRF backbone (reference networks/pdrf/voxnerf.py): tri-plane feature gather + sigma/colour MLPs + TV reg.
//
//   k_points        pts = o + d z                                      (renderer.py:180,206)
//   k_voxel_sample  VoxelNeRFBase.sample / compute_appfeature          (voxnerf.py:203-208,132-151)
//                   planes are kept CHANNEL-LAST on the device ([H][W][C], lines [L][C]) so that one bilinear
//                   tap of one plane is one contiguous 64..256-byte read instead of C strided ones.
//   k_voxel_mlp     VoxelNeRFBase.forward, per-sample part             (voxnerf.py:210-221,240-254)
//                   sigma net + colour net on the same transposed-MFMA machinery as the NeRF backbone.
//   k_tv            TVLoss.forward over one plane/line                 (voxnerf.py:306-324)
#include "mlp_device.h"
#include "voxel.h"
#include "wave_ops.h"

namespace evd {

__global__ void k_points(const float* __restrict__ rb, int nc, const float* __restrict__ z, long n, int S, float* __restrict__ pts) {
    const long s = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (s >= n) return;
    const float* r = rb + (s / S) * nc;
    const float zv = z[s];
#pragma unroll
    for (int c = 0; c < 3; ++c) pts[s * 3 + c] = __fadd_rn(r[c], __fmul_rn(r[3 + c], zv));
}

// merged-sample feature rows of the c2f pass (renderer.py:205-213): out[r, k, 0:F] = order[r, k] < S ? old[r, order] : fresh[r, order - S]
// -- the reference's gather of cat([ft_comb0, ft_comb1]) by the sort order, 16 bytes per lane
__global__ void k_merge_features(const float* __restrict__ old, const float* __restrict__ fresh, const int* __restrict__ order,
                                 long R, int S, int N, int F, float* __restrict__ out, int out_stride) {
    const int per = F / 4;
    const long t = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const long St = S + N;
    if (t >= R * St * per) return;
    const long smp = t / per;
    const int q = t % per;
    const long r = smp / St;
    const int o = order[smp];
    const float* src = o < S ? old + (r * S + o) * (long)F : fresh + (r * N + (o - S)) * (long)F;
    *reinterpret_cast<f32x4*>(out + smp * (long)out_stride + 4 * q) = *reinterpret_cast<const f32x4*>(src + 4 * q);
}

// its backward: the gather is a permutation of the rows of cat([old, fresh]), so every gradient row has exactly one destination
__global__ void k_merge_features_bwd(const float* __restrict__ d_out, int d_stride, const int* __restrict__ order, long R, int S, int N, int F,
                                     float* __restrict__ d_old, float* __restrict__ d_fresh) {
    const int per = F / 4;
    const long t = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const long St = S + N;
    if (t >= R * St * per) return;
    const long smp = t / per;
    const int q = t % per;
    const long r = smp / St;
    const int o = order[smp];
    float* dst = o < S ? d_old + (r * S + o) * (long)F : d_fresh + (r * N + (o - S)) * (long)F;
    *reinterpret_cast<f32x4*>(dst + 4 * q) = *reinterpret_cast<const f32x4*>(d_out + smp * (long)d_stride + 4 * q);
}

__device__ __forceinline__ float unnorm(float c, int size) { return __fmul_rn(__fadd_rn(c, 1.f) / 2.f, (float)(size - 1)); }

constexpr int VS_SAMPLES = 32;      // samples per 256-thread block
constexpr int VS_MAXC = 128;        // max sum(n_comp)

// F.grid_sample(bilinear, zeros, align_corners=True) x 6, product, basis_mat.  The interpolation-weight form
// (w = x - floor x, e = 1 - w) and the tap order follow the ATen CPU kernel, unfused.
//
// Phase 1 (gather): work item = (sample, group of 4 channels); the 4 plane taps and 2 line taps of an item are
// loaded BRANCH-FREE (out-of-range taps read a clamped address and get weight 0 -- the zero padding) and all items of
// a thread are issued ahead of their use, 6-12 independent 8/16-byte loads per lane in flight at 4 wavefronts per SIMD (the grids are
// far larger than L2: this phase is a random gather served by Infinity Cache / HBM).
// Phase 2 (basis_mat, voxnerf.py:151): out^T[f, sample] = basis[f, :] . coef[sample, :] for the block's 32 samples on
// the exact-float32 MFMA (v_mfma_f32_32x32x2_f32 = an fmaf chain in k order), by wavefront 0 of the block.
constexpr int VS_STRIDE = VS_MAXC + 1;      // odd row stride: conflict-free column reads in phase 2
static_assert(VS_SAMPLES * VS_STRIDE >= 3 * 16 * 64 + 32 * 33, "the coefficient array doubles as the reduction buffer + output tile");

struct VsItem {
    f32x4 p[4], l[2];
    float wp[4], wl[2];
    long ip[4], il[2];      // element offsets of the taps (backward: where the gradients are added); unused fields cost the forward nothing
    int grid_id;
    float fw, fn, fl;       // fractional positions inside the cell (x, y of the plane; the line)
    float kx, ky, kl;       // d (pixel coordinate) / d (point coordinate) of the three axes the component reads
    int ax, ay, al;         // ... and which point axes those are
};

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <bool HALF>
__device__ __forceinline__ f32x4 vs_load(const float* base32, const _Float16* base16, long idx) {
    if (HALF) return __builtin_convertvector(*reinterpret_cast<const f16x4*>(base16 + idx), f32x4);
    return *reinterpret_cast<const f32x4*>(base32 + idx);
}

// pick one of three wave-uniform values by a per-lane index WITHOUT indexing the kernel-argument struct dynamically (that
// turns every g.plane[i] / g.grid[..] into a dependent vector load from the argument buffer in front of the real loads)
template <class V> __device__ __forceinline__ V sel3(int i, V a, V b, V c) { return i == 0 ? a : (i == 1 ? b : c); }

template <bool HALF>
__device__ __forceinline__ void vs_issue(const GridParams& g, const float (&pt)[3], int grp, VsItem& it) {
    // matMode = [[0,1],[0,2],[1,2]], vecMode = [2,1,0] (voxnerf.py:99-100)
    int i = 0, c4 = grp * 4;
    if (c4 >= g.n_comp[0]) { c4 -= g.n_comp[0]; i = 1; if (c4 >= g.n_comp[1]) { c4 -= g.n_comp[1]; i = 2; } }
    float xyz[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) xyz[c] = __fsub_rn(__fmul_rn(__fsub_rn(pt[c], g.aabb_min[c]), g.inv[c]), 1.f);   // voxnerf.py:205
    const int C = sel3(i, g.n_comp[0], g.n_comp[1], g.n_comp[2]);
    const int Wp = sel3(i, g.grid[0], g.grid[0], g.grid[1]);          // grid[mat0[i]]
    const int Hp = sel3(i, g.grid[1], g.grid[2], g.grid[2]);          // grid[mat1[i]]
    const int Lp = sel3(i, g.grid[2], g.grid[1], g.grid[0]);          // grid[vec[i]]
    const float cx = sel3(i, xyz[0], xyz[0], xyz[1]), cy = sel3(i, xyz[1], xyz[2], xyz[2]), cl = sel3(i, xyz[2], xyz[1], xyz[0]);
    const float ix = unnorm(cx, Wp), iy = unnorm(cy, Hp);
    // clamp far-away points before the float -> int conversion (everything beyond one cell outside is zero padding)
    const float fx = fminf(fmaxf(floorf(ix), -2.f), (float)Wp), fy = fminf(fmaxf(floorf(iy), -2.f), (float)Hp);
    const float ww = __fsub_rn(ix, floorf(ix)), ee = __fsub_rn(1.f, ww), nn = __fsub_rn(iy, floorf(iy)), ss = __fsub_rn(1.f, nn);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const bool vx0 = x0 >= 0 && x0 < Wp, vx1 = x1 >= 0 && x1 < Wp, vy0 = y0 >= 0 && y0 < Hp, vy1 = y1 >= 0 && y1 < Hp;
    const int cx0 = min(max(x0, 0), Wp - 1), cx1 = min(max(x1, 0), Wp - 1), cy0 = min(max(y0, 0), Hp - 1), cy1 = min(max(y1, 0), Hp - 1);
    const float* pl = sel3(i, g.plane[0], g.plane[1], g.plane[2]) + c4;
    const _Float16* plh = sel3(i, g.plane_h[0], g.plane_h[1], g.plane_h[2]) + c4;
    it.grid_id = i;
    it.fw = ww; it.fn = nn;
    it.kx = 0.5f * (float)(Wp - 1) * sel3(i, g.inv[0], g.inv[0], g.inv[1]);
    it.ky = 0.5f * (float)(Hp - 1) * sel3(i, g.inv[1], g.inv[2], g.inv[2]);
    it.kl = 0.5f * (float)(Lp - 1) * sel3(i, g.inv[2], g.inv[1], g.inv[0]);
    it.ax = sel3(i, 0, 0, 1); it.ay = sel3(i, 1, 2, 2); it.al = sel3(i, 2, 1, 0);
    it.ip[0] = ((long)cy0 * Wp + cx0) * C + c4;
    it.ip[1] = ((long)cy0 * Wp + cx1) * C + c4;
    it.ip[2] = ((long)cy1 * Wp + cx0) * C + c4;
    it.ip[3] = ((long)cy1 * Wp + cx1) * C + c4;
#pragma unroll
    for (int t = 0; t < 4; ++t) it.p[t] = vs_load<HALF>(pl, plh, it.ip[t] - c4);
    it.wp[0] = (vy0 && vx0) ? __fmul_rn(ee, ss) : 0.f;
    it.wp[1] = (vy0 && vx1) ? __fmul_rn(ww, ss) : 0.f;
    it.wp[2] = (vy1 && vx0) ? __fmul_rn(ee, nn) : 0.f;
    it.wp[3] = (vy1 && vx1) ? __fmul_rn(ww, nn) : 0.f;
    const float il = unnorm(cl, Lp);
    const float fl = fminf(fmaxf(floorf(il), -2.f), (float)Lp);
    const float ln = __fsub_rn(il, floorf(il)), ls = __fsub_rn(1.f, ln);
    const int l0 = (int)fl, l1 = l0 + 1;
    const float* li = sel3(i, g.line[0], g.line[1], g.line[2]) + c4;
    const _Float16* lih = sel3(i, g.line_h[0], g.line_h[1], g.line_h[2]) + c4;
    it.fl = ln;
    it.il[0] = (long)min(max(l0, 0), Lp - 1) * C + c4;
    it.il[1] = (long)min(max(l1, 0), Lp - 1) * C + c4;
    it.l[0] = vs_load<HALF>(li, lih, it.il[0] - c4);
    it.l[1] = vs_load<HALF>(li, lih, it.il[1] - c4);
    it.wl[0] = (l0 >= 0 && l0 < Lp) ? ls : 0.f;
    it.wl[1] = (l1 >= 0 && l1 < Lp) ? ln : 0.f;
}

// invalid taps contribute exactly nothing (the reference skips them): a zero weight times a finite grid value is 0,
// and 0 added to the running sum changes nothing
__device__ __forceinline__ f32x4 vs_finish(const VsItem& it, f32x4* pv_out = nullptr, f32x4* lv_out = nullptr) {
    f32x4 pv = {0.f, 0.f, 0.f, 0.f}, lv = {0.f, 0.f, 0.f, 0.f}, cf;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k) pv[k] = it.wp[t] != 0.f ? __fadd_rn(pv[k], __fmul_rn(it.p[t][k], it.wp[t])) : pv[k];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k) lv[k] = it.wl[t] != 0.f ? __fadd_rn(lv[k], __fmul_rn(it.l[t][k], it.wl[t])) : lv[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) cf[k] = __fmul_rn(pv[k], lv[k]);
    if (pv_out) *pv_out = pv;
    if (lv_out) *lv_out = lv;
    return cf;
}

// Tap geometry of one sample in one of its three components (plane i x line i): element offsets of channel 0 and the interpolation
// weights of the 4 plane taps and the 2 line taps.  Computed ONCE per (sample, component) by the forward gather's phase 0 and shared
// through LDS by the component's channel groups -- the first version recomputed it in every (sample, channel group) item: 320 VALU
// instructions per item, the kernel was VALU-bound (PMC: profiles/r02_pmc_voxel.txt).  Same formulas, same order as vs_issue.
struct VsTaps {
    long ip[4], il[2];
    float wp[4], wl[2];
};

__device__ __forceinline__ void vs_geometry(const GridParams& g, const float (&pt)[3], int i, VsTaps& tp) {
    float xyz[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) xyz[c] = __fsub_rn(__fmul_rn(__fsub_rn(pt[c], g.aabb_min[c]), g.inv[c]), 1.f);   // voxnerf.py:205
    const int C = sel3(i, g.n_comp[0], g.n_comp[1], g.n_comp[2]);
    const int Wp = sel3(i, g.grid[0], g.grid[0], g.grid[1]);          // grid[mat0[i]]
    const int Hp = sel3(i, g.grid[1], g.grid[2], g.grid[2]);          // grid[mat1[i]]
    const int Lp = sel3(i, g.grid[2], g.grid[1], g.grid[0]);          // grid[vec[i]]
    const float cx = sel3(i, xyz[0], xyz[0], xyz[1]), cy = sel3(i, xyz[1], xyz[2], xyz[2]), cl = sel3(i, xyz[2], xyz[1], xyz[0]);
    const float ix = unnorm(cx, Wp), iy = unnorm(cy, Hp);
    const float fx = fminf(fmaxf(floorf(ix), -2.f), (float)Wp), fy = fminf(fmaxf(floorf(iy), -2.f), (float)Hp);
    const float ww = __fsub_rn(ix, floorf(ix)), ee = __fsub_rn(1.f, ww), nn = __fsub_rn(iy, floorf(iy)), ss = __fsub_rn(1.f, nn);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const bool vx0 = x0 >= 0 && x0 < Wp, vx1 = x1 >= 0 && x1 < Wp, vy0 = y0 >= 0 && y0 < Hp, vy1 = y1 >= 0 && y1 < Hp;
    const int cx0 = min(max(x0, 0), Wp - 1), cx1 = min(max(x1, 0), Wp - 1), cy0 = min(max(y0, 0), Hp - 1), cy1 = min(max(y1, 0), Hp - 1);
    tp.ip[0] = ((long)cy0 * Wp + cx0) * C;
    tp.ip[1] = ((long)cy0 * Wp + cx1) * C;
    tp.ip[2] = ((long)cy1 * Wp + cx0) * C;
    tp.ip[3] = ((long)cy1 * Wp + cx1) * C;
    tp.wp[0] = (vy0 && vx0) ? __fmul_rn(ee, ss) : 0.f;
    tp.wp[1] = (vy0 && vx1) ? __fmul_rn(ww, ss) : 0.f;
    tp.wp[2] = (vy1 && vx0) ? __fmul_rn(ee, nn) : 0.f;
    tp.wp[3] = (vy1 && vx1) ? __fmul_rn(ww, nn) : 0.f;
    const float il = unnorm(cl, Lp);
    const float fl = fminf(fmaxf(floorf(il), -2.f), (float)Lp);
    const float ln = __fsub_rn(il, floorf(il)), ls = __fsub_rn(1.f, ln);
    const int l0 = (int)fl, l1 = l0 + 1;
    tp.il[0] = (long)min(max(l0, 0), Lp - 1) * C;
    tp.il[1] = (long)min(max(l1, 0), Lp - 1) * C;
    tp.wl[0] = (l0 >= 0 && l0 < Lp) ? ls : 0.f;
    tp.wl[1] = (l1 >= 0 && l1 < Lp) ? ln : 0.f;
}

// GC channels per work item (4, or 8 when every n_comp is a multiple of 8): the gather is bound by the rate at which the texture
// path takes lane addresses (PMC: TCP_TOTAL_CACHE_ACCESSES = one per lane and load; 1171 per wavefront, 300 k cycles per CU), so
// the wider the per-lane load, the fewer of them: 8 float16 channels = one 16-byte load per tap.
#ifdef EVD_VS_TRACE
__device__ long long evd_vs_trace[8 * 8192];
#define VS_STAMP(k) if (threadIdx.x == 0 && blockIdx.x < 8192) evd_vs_trace[blockIdx.x * 8 + (k)] = (long long)__builtin_readcyclecounter()
extern "C" int evd_debug_vs_trace(long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(evd_vs_trace), sizeof(evd_vs_trace)); }
#else
#define VS_STAMP(k)
#endif
#ifndef EVD_VS_WAVES
#define EVD_VS_WAVES 4
#endif
template <bool HALF, int GC>
__global__ __launch_bounds__(256, EVD_VS_WAVES) void k_voxel_sample(const GridParams g, const float* __restrict__ pts, long n,
                                                      float* __restrict__ out, int out_stride, int out_col) {
    __shared__ __attribute__((aligned(16))) float coef[VS_SAMPLES * VS_STRIDE];
    __shared__ __attribute__((aligned(16))) VsTaps taps[VS_SAMPLES * 3];
    constexpr int NV = GC / 4;                   // 4-channel vectors per item
    const int ctot = g.n_comp[0] + g.n_comp[1] + g.n_comp[2];
    const int ng = ctot / GC;
    const long s0 = blockIdx.x * (long)VS_SAMPLES;
    const int items = VS_SAMPLES * ng;
    VS_STAMP(0);
    // phase 2's operand, fetched first so that its latency hides behind the gather: the basis_mat GEMM of the block's 32 samples is
    // split along k over the four wavefronts (a quarter of the components each), every lane keeps its <= 16 basis values in registers
    const bool ksplit = g.app_dim <= 32 && ctot % 8 == 0;
    const int wv = threadIdx.x >> 6, kq = ctot / 4;
    float bq[VS_MAXC / 8];
    if (ksplit) {
        const int lane = threadIdx.x & 63, frow = min(lane & 31, g.app_dim - 1);
        const float* bw = g.basis + (long)frow * ctot + wv * kq + (lane >> 5);
#pragma unroll
        for (int j = 0; j < VS_MAXC / 8; ++j) bq[j] = 2 * j < kq ? bw[2 * j] : 0.f;
    }
    if (threadIdx.x < VS_SAMPLES * 3) {          // phase 0: the tap geometry of every (sample, component) of the block, once
        const int sl = threadIdx.x / 3, i = threadIdx.x % 3;
        const long s = s0 + sl < n ? s0 + sl : n - 1;
        const float pt[3] = {pts[s * 3], pts[s * 3 + 1], pts[s * 3 + 2]};
        VsTaps tp;
        vs_geometry(g, pt, i, tp);
        taps[threadIdx.x] = tp;
    }
    __syncthreads();
    VS_STAMP(1);
    constexpr int UNR = GC == 8 ? 2 : 3;         // n_comp (64,16,16): 32 samples x 12 (24) groups = 1.5 (3) items per thread
    for (int base = threadIdx.x; base < items; base += UNR * 256) {
        VsItem it[UNR][NV];
        int sl[UNR], grp[UNR];
        bool on[UNR];
#pragma unroll
        for (int q = 0; q < UNR; ++q) {             // all tap loads of the thread's items in flight together
            const int t = base + q * 256;
            on[q] = t < items;
            sl[q] = on[q] ? t / ng : 0;
            grp[q] = on[q] ? t % ng : 0;
            int i = 0, c4 = grp[q] * GC;
            if (c4 >= g.n_comp[0]) { c4 -= g.n_comp[0]; i = 1; if (c4 >= g.n_comp[1]) { c4 -= g.n_comp[1]; i = 2; } }
            const VsTaps& tp = taps[sl[q] * 3 + i];
            const float* pl = sel3(i, g.plane[0], g.plane[1], g.plane[2]) + c4;
            const _Float16* plh = sel3(i, g.plane_h[0], g.plane_h[1], g.plane_h[2]) + c4;
            const float* li = sel3(i, g.line[0], g.line[1], g.line[2]) + c4;
            const _Float16* lih = sel3(i, g.line_h[0], g.line_h[1], g.line_h[2]) + c4;
#ifdef EVD_VS_ABL_NOLOAD
            const long zero_ = (long)(threadIdx.x & 0);
#define VS_IDX(x) (zero_ + ((x) & 0))
#else
#define VS_IDX(x) (x)
#endif
            if (HALF && GC == 8) {                  // one 16-byte load per tap
                typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f16x8v v = *reinterpret_cast<const f16x8v*>(plh + VS_IDX(tp.ip[k]));
#pragma unroll
                    for (int e = 0; e < 8; ++e) it[q][e >> 2].p[k][e & 3] = (float)v[e];
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const f16x8v v = *reinterpret_cast<const f16x8v*>(lih + VS_IDX(tp.il[k]));
#pragma unroll
                    for (int e = 0; e < 8; ++e) it[q][e >> 2].l[k][e & 3] = (float)v[e];
                }
            } else {
#pragma unroll
                for (int v = 0; v < NV; ++v) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) it[q][v].p[k] = vs_load<HALF>(pl, plh, tp.ip[k] + 4 * v);
#pragma unroll
                    for (int k = 0; k < 2; ++k) it[q][v].l[k] = vs_load<HALF>(li, lih, tp.il[k] + 4 * v);
                }
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) {
#pragma unroll
                for (int k = 0; k < 4; ++k) it[q][v].wp[k] = tp.wp[k];
#pragma unroll
                for (int k = 0; k < 2; ++k) it[q][v].wl[k] = tp.wl[k];
            }
        }
#pragma unroll
        for (int q = 0; q < UNR; ++q) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const f32x4 cf = vs_finish(it[q][v]);
                if (on[q]) {
                    float* dst = &coef[sl[q] * VS_STRIDE + grp[q] * GC + 4 * v];      // odd row stride: scalar stores
#pragma unroll
                    for (int k = 0; k < 4; ++k) dst[k] = cf[k];
                }
            }
        }
    }
    VS_STAMP(2);
    __syncthreads();
    VS_STAMP(3);
#ifdef EVD_VS_ABL_NOPHASE2
    if (threadIdx.x < 32 && s0 + threadIdx.x < n) out[(s0 + threadIdx.x) * (long)out_stride + out_col] = coef[threadIdx.x * VS_STRIDE];
    return;
#endif
    if (ksplit) {
        const int lane = threadIdx.x & 63, col = lane & 31, hh = lane >> 5;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* cf = coef + col * VS_STRIDE + wv * kq + hh;
#pragma unroll
        for (int j = 0; j < VS_MAXC / 8; ++j)
            if (2 * j < kq) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[j], cf[2 * j], acc, 0, 0, 0);
        VS_STAMP(4);
        __syncthreads();                         // every wavefront has read its coefficients: the array becomes the reduction buffer
        if (wv > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) coef[((wv - 1) * 16 + r) * 64 + lane] = acc[r];
        }
        __syncthreads();
        // wavefront 0 sums the four partial tiles, applies the activation and transposes the tile through LDS; then ALL threads store:
        // a lane per (sample, feature), 128-byte runs per sample row (the rows of the level's input matrix are 380 / 508 bytes apart:
        // stored straight from the accumulator layout they were 16.8 M scattered 4-byte writes per launch -- 100 us of a 160 us kernel)
        float* ot = coef + 3 * 16 * 64;           // [32 samples][33]
        if (wv == 0) {
#pragma unroll
            for (int w = 0; w < 3; ++w)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] += coef[(w * 16 + r) * 64 + lane];
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[col * 33 + (r & 3) + 8 * (r >> 2) + 4 * hh] = act(g.app_act, acc[r]);
        }
        VS_STAMP(5);
        __syncthreads();
        VS_STAMP(6);
        for (int t = threadIdx.x; t < VS_SAMPLES * 32; t += 256) {
            const int sl = t >> 5, f = t & 31;
            if (s0 + sl < n && f < g.app_dim) out[(s0 + sl) * (long)out_stride + out_col + f] = ot[sl * 33 + f];
        }
        VS_STAMP(7);
        return;
    }
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x, col = lane & 31, hh = lane >> 5;
    for (int f0 = 0; f0 < g.app_dim; f0 += 32) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const int frow = min(f0 + col, g.app_dim - 1);
        const float* bw = g.basis + (long)frow * ctot + hh;
        const float* cf = coef + col * VS_STRIDE + hh;
        for (int kk = 0; kk < ctot; kk += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bw[kk], cf[kk], acc, 0, 0, 0);
        const long s = s0 + col;
        if (s < n) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int f = f0 + 8 * q + 4 * hh;
                float* o = out + s * (long)out_stride + out_col + f;
                if (f + 3 < g.app_dim && ((out_stride | out_col) & 3) == 0) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act(g.app_act, acc[4 * q + e]);
                    *reinterpret_cast<f32x4*>(o) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (f + e < g.app_dim) o[e] = act(g.app_act, acc[4 * q + e]);
                }
            }
        }
    }
}

// Wavefront-autonomous form of the gather (the one the shipped levels run: every n_comp a multiple of 8, app_dim <= 32).
// PMC + in-kernel stamps of the block-cooperative kernel above (profiles/r02_pmc_voxel.txt): it is neither bandwidth- nor VALU-bound
// but a chain of latencies separated by block barriers (points -> geometry | barrier | gather | barrier | basis GEMM | barrier | reduce |
// barrier | store: 22 k cycles per 32 samples, 4 blocks per CU) -- with L2-resident toy grids it runs at the same speed.  Here a
// WAVEFRONT owns 16 samples from the point load to the store and never waits for another wavefront: geometry of its 48 (sample,
// component) pairs on 48 lanes -> its own LDS slice -> 3 items per lane (16 samples x 12 groups of 8 channels, 18 16-byte loads in
// flight) -> coefficients in LDS -> out^T = basis . coef^T on v_mfma_f32_16x16x4_f32 (2 feature tiles x ctot / 4 steps) -> transposed
// through LDS -> 128-byte runs per sample row.  The 4 wavefronts of a SIMD run their chains independently, so one wavefront's
// matrix work and stores overlap the others' gathers.  (Measured and dropped: persistent wavefronts walking 8 sample groups each so
// that the block's basis_mat load is paid once -- 0.539 / 0.516 / 0.530 ms per c2f render with 1024 / 2048 / 512 blocks against
// 0.521 ms for one group per wavefront: the other wavefronts already hide that prologue.)
constexpr int VW_SAMPLES = 16;                  // samples per wavefront
constexpr int VW_WAVES = 4;                     // wavefronts per block
// LDS slice of one wavefront: tap table, then the coefficient rows [16][ctot + 1] (later the output tile [16][33])
__host__ __device__ constexpr size_t vw_basis_bytes(int ctot) { return (size_t)32 * (ctot + 1) * 4 + 16 - ((size_t)32 * (ctot + 1) * 4) % 16; }
constexpr int VW_OS = 36;                       // row stride of the output tile in LDS (floats): 16-byte aligned rows
__host__ __device__ constexpr size_t vw_slice_bytes(int ctot) { return ((VW_SAMPLES * 3 * sizeof(VsTaps) + (size_t)VW_SAMPLES * (ctot + 1 > VW_OS ? ctot + 1 : VW_OS) * 4) + 15) & ~(size_t)15; }
// OCC: wavefronts per SIMD the kernel is compiled for (registers <= 512 / OCC)
template <bool HALF, int OCC>
__global__ __launch_bounds__(64 * VW_WAVES, OCC) void k_voxel_sample_w(const GridParams g, const float* __restrict__ pts, long n,
                                                                float* __restrict__ out, int out_stride, int out_col) {
    extern __shared__ __attribute__((aligned(16))) char vw_smem[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ctot = g.n_comp[0] + g.n_comp[1] + g.n_comp[2];
    const int cstride = ctot + 1;                // odd row stride (ctot is a multiple of 8): conflict-free column reads of the GEMM
    float* bs = reinterpret_cast<float*>(vw_smem);                       // basis_mat [32][ctot + 1], shared by the block
    char* slice = vw_smem + vw_basis_bytes(ctot) + (size_t)wv * vw_slice_bytes(ctot);
    VsTaps* taps = reinterpret_cast<VsTaps*>(slice);
    float* coef = reinterpret_cast<float*>(slice + VW_SAMPLES * 3 * sizeof(VsTaps));
    const int ng = ctot / 8;
    const long s0 = ((long)blockIdx.x * VW_WAVES + wv) * VW_SAMPLES;
    // basis_mat -> LDS: the loads are issued first and land while the geometry is computed
    constexpr int NBV = (32 * VS_MAXC / 4 + 64 * VW_WAVES - 1) / (64 * VW_WAVES);
    f32x4 bv[NBV];
    const int nb4 = g.app_dim * ctot / 4;
#pragma unroll
    for (int q = 0; q < NBV; ++q) {
        const int i4 = threadIdx.x + q * 64 * VW_WAVES;
        bv[q] = i4 < nb4 ? *reinterpret_cast<const f32x4*>(g.basis + 4 * i4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    VS_STAMP(0);
    if (s0 < n && lane < VW_SAMPLES * 3) {       // geometry of this wavefront's (sample, component) pairs
        const int sl = lane / 3, i = lane % 3;
        const long s = s0 + sl < n ? s0 + sl : n - 1;
        const float pt[3] = {pts[s * 3], pts[s * 3 + 1], pts[s * 3 + 2]};
        VsTaps tp;
        vs_geometry(g, pt, i, tp);
        taps[lane] = tp;
    }
#pragma unroll
    for (int q = 0; q < NBV; ++q) {
        const int i4 = threadIdx.x + q * 64 * VW_WAVES;
        if (i4 < nb4) {
            const int f = (4 * i4) / ctot, c = (4 * i4) % ctot;      // ctot is a multiple of 4: the 4 values stay in one row
#pragma unroll
            for (int e = 0; e < 4; ++e) bs[f * cstride + c + e] = bv[q][e];
        }
    }
    __syncthreads();                             // the only block-wide barrier: basis_mat visible (also orders the tap tables)
    if (s0 >= n) return;
    VS_STAMP(1);
    const int items = VW_SAMPLES * ng;
    typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
    constexpr int UNR = 3;                       // 16 samples x 12 groups = 3 items per lane
    constexpr int NRAW = HALF ? 1 : 2;           // 16-byte loads per tap
    for (int base = lane; base < items; base += UNR * 64) {
        f32x4 rawp[UNR][4][NRAW], rawl[UNR][2][NRAW];    // the taps as loaded (float16 x 8 in one f32x4 register quad, or 2 x float32 x 4)
        int sl[UNR], grp[UNR], comp[UNR];
        bool on[UNR];
#pragma unroll
        for (int q = 0; q < UNR; ++q) {
            const int t = base + q * 64;
            on[q] = t < items;
            sl[q] = on[q] ? t / ng : 0;
            grp[q] = on[q] ? t % ng : 0;
            int i = 0, c8 = grp[q] * 8;
            if (c8 >= g.n_comp[0]) { c8 -= g.n_comp[0]; i = 1; if (c8 >= g.n_comp[1]) { c8 -= g.n_comp[1]; i = 2; } }
            comp[q] = i;
            const VsTaps& tp = taps[sl[q] * 3 + i];
#ifdef EVD_VS_NO_LOADS        // developer ablation (tools/dev/gather_ablation.sh): everything but the grid loads (values made from the tap offsets)
            if (true) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int v = 0; v < NRAW; ++v) rawp[q][k][v] = f32x4{(float)tp.ip[k], 1.f, 2.f, (float)c8};
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int v = 0; v < NRAW; ++v) rawl[q][k][v] = f32x4{(float)tp.il[k], 1.f, 2.f, (float)c8};
            } else
#endif
            if (HALF) {
                const _Float16* plh = sel3(i, g.plane_h[0], g.plane_h[1], g.plane_h[2]) + c8;
                const _Float16* lih = sel3(i, g.line_h[0], g.line_h[1], g.line_h[2]) + c8;
#pragma unroll
                for (int k = 0; k < 4; ++k) rawp[q][k][0] = *reinterpret_cast<const f32x4*>(plh + tp.ip[k]);
#pragma unroll
                for (int k = 0; k < 2; ++k) rawl[q][k][0] = *reinterpret_cast<const f32x4*>(lih + tp.il[k]);
            } else {
                const float* pl = sel3(i, g.plane[0], g.plane[1], g.plane[2]) + c8;
                const float* li = sel3(i, g.line[0], g.line[1], g.line[2]) + c8;
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int v = 0; v < NRAW; ++v) rawp[q][k][v] = *reinterpret_cast<const f32x4*>(pl + tp.ip[k] + 4 * v);
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int v = 0; v < NRAW; ++v) rawl[q][k][v] = *reinterpret_cast<const f32x4*>(li + tp.il[k] + 4 * v);
            }
        }
#pragma unroll
        for (int q = 0; q < UNR; ++q) {
            const VsTaps& tp = taps[sl[q] * 3 + comp[q]];
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                VsItem it;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (HALF) {
                        const f16x8v h8 = __builtin_bit_cast(f16x8v, rawp[q][k][0]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) it.p[k][e] = (float)h8[4 * v + e];
                    } else {
                        it.p[k] = rawp[q][k][HALF ? 0 : v];
                    }
                    it.wp[k] = tp.wp[k];
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    if (HALF) {
                        const f16x8v h8 = __builtin_bit_cast(f16x8v, rawl[q][k][0]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) it.l[k][e] = (float)h8[4 * v + e];
                    } else {
                        it.l[k] = rawl[q][k][HALF ? 0 : v];
                    }
                    it.wl[k] = tp.wl[k];
                }
                const f32x4 cf = vs_finish(it);
                if (on[q]) {
                    float* dst = &coef[sl[q] * cstride + grp[q] * 8 + 4 * v];
#pragma unroll
                    for (int k = 0; k < 4; ++k) dst[k] = cf[k];
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // this wavefront's LDS writes before its own reads: program order
    __builtin_amdgcn_wave_barrier();
    VS_STAMP(2);
    // out^T[f, sample] = sum_k basis[f, k] coef[sample, k]:  D lane l, reg r = feature 16 tile + 4 (l / 16) + r, sample l % 16
    const int col = lane & 15, kh = lane >> 4;
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const float* b0 = bs + min(col, g.app_dim - 1) * cstride + kh;
    const float* b1 = bs + min(16 + col, g.app_dim - 1) * cstride + kh;
    const float* cf = coef + col * cstride + kh;
    // The k loop runs in groups of four steps (ctot is a multiple of 8; a last half group where it is not one of 16): the twelve LDS
    // operands of the NEXT group are read before the eight MFMAs of the current one are issued, so the matrix core never waits for a
    // ds_read (in-kernel stamps, 16 samples: 4.2 k cycles for this phase with every step waiting for its own three reads).
    {
        float pb0[4], pb1[4], pc[4];
        auto fetch = [&](int kk, int cnt) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < cnt) { pb0[j] = b0[kk + 4 * j]; pb1[j] = b1[kk + 4 * j]; pc[j] = cf[kk + 4 * j]; }
        };
        int kk = 0;
        fetch(0, ctot >= 16 ? 4 : ctot / 4);
        for (; kk + 16 <= ctot; kk += 16) {
            float cb0[4], cb1[4], cc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { cb0[j] = pb0[j]; cb1[j] = pb1[j]; cc[j] = pc[j]; }
            const int left = ctot - (kk + 16);
            if (left > 0) fetch(kk + 16, left >= 16 ? 4 : left / 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(cb0[j], cc[j], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(cb1[j], cc[j], acc[1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j)                          // the half group (its operands are in the prefetch registers)
            if (kk + 4 * j < ctot) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(pb0[j], pc[j], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(pb1[j], pc[j], acc[1], 0, 0, 0);
            }
    }
    VS_STAMP(3);
    __builtin_amdgcn_wave_barrier();             // every lane has read its coefficients: the slice becomes the output tile [16][VW_OS]
#pragma unroll
    for (int tile = 0; tile < 2; ++tile) {
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (16 * tile + 4 * kh + r) < g.app_dim ? act(g.app_act, acc[tile][r]) : 0.f;
        *reinterpret_cast<f32x4*>(&coef[col * VW_OS + 16 * tile + 4 * kh]) = v;      // rows of 36 floats: 16-byte aligned
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (g.app_dim == 32) {                        // a lane stores 8 consecutive features of a sample: 128-byte runs per sample row, two 16-byte
        const int sl = lane >> 2, f0 = 8 * (lane & 3);      // stores per lane (the rows of the level's input matrix are only 4-byte aligned)
        if (s0 + sl < n) {
            typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
            const f32x4 a = *reinterpret_cast<const f32x4*>(&coef[sl * VW_OS + f0]), b = *reinterpret_cast<const f32x4*>(&coef[sl * VW_OS + f0 + 4]);
            float* o = out + (s0 + sl) * (long)out_stride + out_col + f0;
            *reinterpret_cast<f32x4u*>(o) = a;
            *reinterpret_cast<f32x4u*>(o + 4) = b;
        }
    } else {
        for (int t = lane; t < VW_SAMPLES * 32; t += 64) {
            const int sl = t >> 5, f = t & 31;
            if (s0 + sl < n && f < g.app_dim) out[(s0 + sl) * (long)out_stride + out_col + f] = coef[sl * VW_OS + f];
        }
    }
    VS_STAMP(4);
    VS_STAMP(5); VS_STAMP(6); VS_STAMP(7);
}

typedef float vbw_f32x2 __attribute__((ext_vector_type(2)));
// Power of two that brings a magnitude m into [2^13, 2^14) (float16's largest binades, so that hi / lo splits of values up to m keep
// 2^-22 of m), and its inverse; m = 0, denormal or tiny: the scale of 2^-113; non-finite m passes through (the scaled values are then
// non-finite as well and so is the product, as in float32).
__device__ __forceinline__ float vbw_pow2_scale(float m, float* inv) {
    int E = (int)((__float_as_uint(m) >> 23) & 0xffu);
    E = E < 14 ? 14 : E;
    *inv = __uint_as_float((unsigned)(E - 13) << 23);
    return __uint_as_float((unsigned)(267 - E) << 23);
}
typedef unsigned vbw_u32x4 __attribute__((ext_vector_type(4)));
// acc + w x (one float16 of a packed pair) in ONE instruction: v_fma_mix_f32 (op_sel_hi marks the float16 source, op_sel picks its high half);
// hipcc does not form it from fmaf(w, (float)h, acc) here (a conversion + a fused multiply-add: 48 more instructions per gathered item)
template <int HI> __device__ __forceinline__ float vbw_fma_mix(float w, unsigned pair, float acc) {
    float d;
    if (HI) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(d) : "v"(w), "v"(pair), "v"(acc));
    else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,1,0]" : "=v"(d) : "v"(w), "v"(pair), "v"(acc));
    return d;
}
__device__ __forceinline__ float vbw_mul_legacy(float a, float b) {          // a x b with 0 x anything = 0 (v_mul_legacy_f32: VOP3 only, no builtin in this hipcc)
    float d;
    asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// ------------------------------------------------------------------------------------------------------------------------------------------------
// k_voxel_sample_m (round 6): the gather rebuilt around what actually bounds it.  An ablation build of k_voxel_sample_w WITHOUT its grid loads
// runs at 70.1 us against 74.0 us with them (profiles/r06_gather_ablation.log): the kernel was never bound by the gather -- it issues ~3000
// vector instructions per wavefront and 16 samples (static count of its ISA: interpolation with compare / select per element, float16 ->
// float32 conversions, 64-bit tap offsets, the coefficient round trip through LDS for a float32 16 x 16 x 4 GEMM of 48 MFMAs with three LDS
// reads each, the output transposed through LDS), and at four wavefronts per SIMD that IS its duration.  Here:
//   * a lane's work item is (sample = lane % 16, 8-channel group = 4 q + lane / 16), q = 0 .. ctot / 32 - 1: exactly the B-operand layout of
//     v_mfma_f32_16x16x32_f16 (lane holds k = 8 (lane / 16) .. + 7 of column lane % 16) -- the eight coefficients a lane computes ARE its
//     operand, nothing goes through LDS;
//   * out^T[f, sample] = sum_k basis[f, k] coef[sample, k] on the float16 matrix core in the split form the float32-grade modes use everywhere
//     (hi = f16(x), lo = f16(x - hi); A_hi B_hi + A_hi B_lo + A_lo B_hi, float32 accumulate: 2^-21 relative per product): 18 MFMAs of 16 cycles
//     instead of 48 of 32; the split basis_mat operands are made once per workgroup in LDS (12 KiB), the workgroups are persistent;
//   * the D layout (feature 4 (lane / 16) + r of sample lane % 16) is four consecutive floats of a sample's output row: stored straight from the
//     accumulators, no transposition;
//   * 32-bit tap offsets; HALF: weight x float16 value + sum as ONE v_fma_mix_f32 (no conversion, no separate multiply; the float16 grids are
//     finite, so a zero weight needs no guard); float32 grids: v_mul_legacy_f32 + add (the guarded sum of rounds 1-5, exactly).
struct VmTaps { int ip[4], il[2]; float wp[4], wl[2]; };          // 48 bytes: element offsets of channel 0 (clamped) and weights (0 = outside)
constexpr int VM_WAVES = 4;
typedef _Float16 vm_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 vm_h2 __attribute__((ext_vector_type(2)));
typedef float vm_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void vm_geometry(const GridParams& g, const float (&pt)[3], int i, VmTaps& tp) {       // vs_geometry, 32-bit offsets
    VsTaps t;
    vs_geometry(g, pt, i, t);
#pragma unroll
    for (int k = 0; k < 4; ++k) { tp.ip[k] = (int)t.ip[k]; tp.wp[k] = t.wp[k]; }
#pragma unroll
    for (int k = 0; k < 2; ++k) { tp.il[k] = (int)t.il[k]; tp.wl[k] = t.wl[k]; }
}

template <bool HALF, int OCC>
__global__ __launch_bounds__(64 * VM_WAVES, OCC) void k_voxel_sample_m(const GridParams g, const float* __restrict__ pts, long n,
                                                                       float* __restrict__ out, int out_stride, int out_col) {
    __shared__ __attribute__((aligned(16))) vm_h8 a_hi[2 * 3 * 64], a_lo[2 * 3 * 64];        // [feature tile][k step][lane]
    __shared__ __attribute__((aligned(16))) VmTaps taps_all[VM_WAVES][16 * 3];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c0n = g.n_comp[0], c1n = g.n_comp[1], ctot = c0n + c1n + g.n_comp[2], steps = ctot / 32, F = g.app_dim;
    // the split A operands: entry (tile, step, lane) = basis[16 tile + lane % 16][32 step + 8 (lane / 16) .. + 7]
    for (int e = threadIdx.x; e < 2 * 3 * 64; e += 64 * VM_WAVES) {
        const int l = e & 63, st = (e >> 6) % 3, tl = e / 192, f = 16 * tl + (l & 15), k0 = 32 * st + 8 * (l >> 4);
        vm_h8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = (f < F && st < steps) ? g.basis[(long)f * ctot + k0 + j] : 0.f;
            hi[j] = (_Float16)v;
            lo[j] = (_Float16)(v - (float)hi[j]);
        }
        a_hi[e] = hi;
        a_lo[e] = lo;
    }
    __syncthreads();
    VmTaps* taps = taps_all[wv];
    const int col = lane & 15, kb = lane >> 4;
    const long tiles = (n + 15) / 16;
    for (long tile = (long)blockIdx.x * VM_WAVES + wv; tile < tiles; tile += (long)gridDim.x * VM_WAVES) {
        const long s0 = tile * 16;
        VS_STAMP(0);
        // (measured and dropped: the NEXT tile's points fetched here, one tile ahead -- 57.8 vs 56.7 us: their latency is not what the tile waits for)
        if (lane < 48) {                              // geometry of this wavefront's (sample, component) pairs, once each
            const int sl = lane / 3, i = lane % 3;
            const long s = s0 + sl < n ? s0 + sl : n - 1;
            const float pt[3] = {pts[s * 3], pts[s * 3 + 1], pts[s * 3 + 2]};
            VmTaps tp;
            vm_geometry(g, pt, i, tp);
            taps[lane] = tp;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        VS_STAMP(1);
        constexpr int NRAW = HALF ? 1 : 2;
        f32x4 rawp[3][4][NRAW], rawl[3][2][NRAW];
        int tix[3];                                   // the item's row of the tap table: the weights are read again when the values have landed
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if (q < steps) {
                int c8 = 32 * q + 8 * kb, i = 0;
                if (c8 >= c0n) { c8 -= c0n; i = 1; if (c8 >= c1n) { c8 -= c1n; i = 2; } }
                tix[q] = col * 3 + i;
                struct { int ip[4], il[2]; } tq[3];
#pragma unroll
                for (int k = 0; k < 4; ++k) tq[q].ip[k] = taps[tix[q]].ip[k];
#pragma unroll
                for (int k = 0; k < 2; ++k) tq[q].il[k] = taps[tix[q]].il[k];
                if (HALF) {
                    const _Float16* plh = sel3(i, g.plane_h[0], g.plane_h[1], g.plane_h[2]) + c8;
                    const _Float16* lih = sel3(i, g.line_h[0], g.line_h[1], g.line_h[2]) + c8;
#pragma unroll
                    for (int k = 0; k < 4; ++k) rawp[q][k][0] = *reinterpret_cast<const f32x4*>(plh + tq[q].ip[k]);
#pragma unroll
                    for (int k = 0; k < 2; ++k) rawl[q][k][0] = *reinterpret_cast<const f32x4*>(lih + tq[q].il[k]);
                } else {
                    const float* pl = sel3(i, g.plane[0], g.plane[1], g.plane[2]) + c8;
                    const float* li = sel3(i, g.line[0], g.line[1], g.line[2]) + c8;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int v = 0; v < NRAW; ++v) rawp[q][k][v] = *reinterpret_cast<const f32x4*>(pl + tq[q].ip[k] + 4 * v);
#pragma unroll
                    for (int k = 0; k < 2; ++k)
#pragma unroll
                        for (int v = 0; v < NRAW; ++v) rawl[q][k][v] = *reinterpret_cast<const f32x4*>(li + tq[q].il[k] + 4 * v);
                }
            }
        }
        VS_STAMP(2);
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if (q < steps) {
                float cf[8];
                struct { float wp[4], wl[2]; } tq[3];
#pragma unroll
                for (int k = 0; k < 4; ++k) tq[q].wp[k] = taps[tix[q]].wp[k];
#pragma unroll
                for (int k = 0; k < 2; ++k) tq[q].wl[k] = taps[tix[q]].wl[k];
                if (HALF) {
                    float pv[8], lv[8];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const vm_h8 h = __builtin_bit_cast(vm_h8, rawp[q][k][0]);
#pragma unroll
                        for (int e = 0; e < 8; ++e) pv[e] = k == 0 ? (float)h[e] * tq[q].wp[0] : __builtin_fmaf((float)h[e], tq[q].wp[k], pv[e]);
                    }
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const vm_h8 h = __builtin_bit_cast(vm_h8, rawl[q][k][0]);
#pragma unroll
                        for (int e = 0; e < 8; ++e) lv[e] = k == 0 ? (float)h[e] * tq[q].wl[0] : __builtin_fmaf((float)h[e], tq[q].wl[k], lv[e]);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) cf[e] = pv[e] * lv[e];
                } else {
#pragma unroll
                    for (int v = 0; v < 2; ++v) {
                        f32x4 pv = {0.f, 0.f, 0.f, 0.f}, lv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int k = 0; k < 4; ++k)
#pragma unroll
                            for (int e = 0; e < 4; ++e) pv[e] = __fadd_rn(pv[e], vbw_mul_legacy(tq[q].wp[k], rawp[q][k][v][e]));
#pragma unroll
                        for (int k = 0; k < 2; ++k)
#pragma unroll
                            for (int e = 0; e < 4; ++e) lv[e] = __fadd_rn(lv[e], vbw_mul_legacy(tq[q].wl[k], rawl[q][k][v][e]));
#pragma unroll
                        for (int e = 0; e < 4; ++e) cf[4 * v + e] = __fmul_rn(pv[e], lv[e]);
                    }
                }
                vm_h8 bh, bl;
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const vm_h2 h2 = __builtin_convertvector(vm_f2{cf[e], cf[e + 1]}, vm_h2);
                    const vm_h2 l2 = __builtin_convertvector(vm_f2{cf[e] - (float)h2[0], cf[e + 1] - (float)h2[1]}, vm_h2);
                    bh[e] = h2[0]; bh[e + 1] = h2[1];
                    bl[e] = l2[0]; bl[e + 1] = l2[1];
                }
#pragma unroll
                for (int tl = 0; tl < 2; ++tl) {
                    const vm_h8 ah = a_hi[(tl * 3 + q) * 64 + lane], al = a_lo[(tl * 3 + q) * 64 + lane];
                    acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[tl], 0, 0, 0);
                    acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[tl], 0, 0, 0);
                    acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[tl], 0, 0, 0);
                }
#ifdef EVD_VM_SERIAL
                __builtin_amdgcn_sched_barrier(0);        // one item after the other: the next item's values stay in their load registers
#endif
            }
        }
        VS_STAMP(3);
        // D: lane (col = sample, kb), register r = feature 16 tl + 4 kb + r: four consecutive floats of the sample's output row
        if (s0 + col < n) {
            typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
            float* o = out + (s0 + col) * (long)out_stride + out_col;
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) {
                const int f0 = 16 * tl + 4 * kb;
                f32x4 v = acc[tl];
                if (g.app_act != EVD_ACT_NONE) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = act(g.app_act, acc[tl][r]);
                }
                if (f0 + 3 < F) *reinterpret_cast<f32x4u*>(o + f0) = v;
                else
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (f0 + r < F) o[f0 + r] = v[r];
            }
        }
        VS_STAMP(4);
        __builtin_amdgcn_wave_barrier();              // the tap table is rewritten by the next tile
    }
}

// Backward of k_voxel_sample (app_act none): d out [n, app_dim] -> gradients of the planes, lines (scatter-add, the transpose of
// the gather: the same 4 + 2 taps with the same weights) and of basis_mat.  Persistent blocks walk 32-sample tiles; inside a
// tile the LANES RUN OVER CHANNELS (the grids are channel-last), so every gather and every atomic of a wavefront covers
// contiguous 64..256-byte runs of one tap:
//   A  d out rows -> LDS;  d coef[s, c] = sum_f d out[s, f] basis[f, c];  tap table (18 per sample: 3 x 4 plane + 3 x 2 line)
//   B  plane value pv[s, c], line value lv[s, c] (coalesced gathers), then
//      d plane[tap, c] += w_tap d coef lv,  d line[tap, c] += w_tap d coef pv  as hardware float32 atomics (global_atomic_add_f32);
//      like the reference's grid_sample backward (voxnerf.py:144) the summation order, hence the last bits, is not deterministic
//   C  d basis[f, c] += sum_s d out[s, f] pv lv in registers across tiles, one atomic flush per block at the end
// Measured (fine level, 2^19 samples, 302 M float atomics): **1.23 ms = 246 G adds/s**, the hardware rate of one dword per clock per L2
// channel (128 channels).  Round 1: 1.45 ms (1.80 ms with the two small GEMMs on the VALU).  Round 2: the kernel compiled to 256 VGPRs
// + 109 AGPRs under a loose launch bound, i.e. ONE block per CU, and its non-atomic work (0.98 ms: per-tile latency chain of point
// load, tap table, GEMM, gathers) barely hid under the atomics; with the VALU fallback's accumulators templated out (MM) and
// __launch_bounds__(256, 2) it takes 172 VGPRs, two blocks per CU share the latency, and the gather sweep is unrolled 4 x:
// 1.43 -> 1.23 ms, whole blurfactory iteration 32.6 -> 28.9 ms.  Plane-only and line-only variants
// cost the same per add, and 32 private copies of the (heavily shared) line gradients change nothing: it is the op count, not
// contention.  Tried and dropped: a run-length sum over the tile's consecutive samples that hit the same cell before the atomic (one
// thread per (tap, channel) walking the 32 samples): the sequential walk costs more than the adds it saves (3.1 ms) unless the rays
// run along a grid axis.  Also tried and dropped (round 2): per-tile LDS windows (8 x 8 plane cells / 32 line cells around the tile's
// taps, ds_add_f32, one global atomic per touched cell) for the components whose taps stay together along a ray -- the bounding-box
// atomics, the per-tap window index and the flush add ~0.5 ms per 2^19 samples to this one-wavefront-per-SIMD kernel and the whole
// blurfactory iteration went from 33.2 to 42.1 ms; and the sort + LDS-tile form of kernel_voxel_scatter.hip (2 x slower as built).
// And, once ds_add_f32 was known to be the slow part (kernel_voxel_scatter.hip), the same window for the 64-channel x-y plane WITHOUT
// atomics (lane = channel, a window cell owned by one wavefront, every wavefront walks the tile's 128 (sample, tap) entries in order;
// the box of the benchmark's NDC rays is 42-60 cells, 22-28 of them touched by the 128 taps): correct, and 0.92 -> 1.43 ms per 2^19
// samples, iteration 22.6 -> 30.3 ms -- the walk is a chain of dependent LDS read-modify-writes, again.
// BINNED (kernel_voxel_scatter.hip): the atomics of phase B are replaced by one row of per-channel contributions per sample and the
// tap records; a second pass adds them tile by tile in LDS.  d basis and d pts are computed here either way.
constexpr int VSB_MAXF = 64, VSB_TAPS = 18;
#ifdef EVD_VSB_BATCH
constexpr int VSB_BATCH = EVD_VSB_BATCH;
#else
constexpr int VSB_BATCH = 4;            // samples whose taps are in flight together in the gather phase of k_voxel_sample_bwd (divides 16)
#endif
// CT: the channel capacity the LDS rows are laid out for (MM: ctot <= CT, a multiple of 32).  With the shipped 96 channels and
// app_dim 32 the block needs 50 KB of LDS and 168 VGPRs = three blocks per CU.  (Measured: three blocks run at the speed of two,
// 1.21 ms = 249 G adds/s; a bare kernel of coalesced float atomics on random 64-byte runs sustains 318 - 328 G adds/s = 20 G requests/s
// regardless of the table size, tools/probes/atomic_probe.hip, and the counters show EVERY atomic request of this kernel travelling to
// the memory side, TCC_EA0_ATOMIC == TCC_ATOMIC = 18.6 M 64-byte requests per 2^19 samples: device-scope float atomics are not
// executed in the XCD's L2.  The kernel is at 78 % of that ceiling; the gap is the repeated hits on the same few line cells.)
// sum over the 16 lanes of a DPP row, left in every lane of the row (lanes that are switched off contribute 0)
__device__ __forceinline__ float row_sum_dpp(float v) {
    auto d = [](float src, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, src), decltype(ctrl)::value, 0xf, 0xf, false));
    };
    v += d(v, std::integral_constant<int, 0xb1>());      // quad_perm [1,0,3,2]
    v += d(v, std::integral_constant<int, 0x4e>());      // quad_perm [2,3,0,1]
    v += d(v, std::integral_constant<int, 0x141>());     // row_half_mirror
    v += d(v, std::integral_constant<int, 0x140>());     // row_mirror
    return v;
}

// MODE: 0 = every tap by a direct atomic; 1 = pass 1 of the binned form (rows + tap records + keys, no atomics); 2 = HYBRID: the plane
// taps by direct atomics, the line taps deferred -- their rows and tap records are written and k_scatter_lines adds them through
// privatised LDS slices of the (small) line gradients: a third of the kernel's atomic requests go away.
// MODE 3 (opt-in, EVD_SCATTER_WIN=1) = the hybrid form with the x-y plane's taps deferred as well (k_scatter_xy, kernel_voxel_scatter.hip):
// their rows (64 channels) and tap records are written here and this kernel's atomics are the two 16-channel planes only.  Measured at 2^19
// fine-level samples: this kernel 0.89 -> 0.76 ms although two thirds of its atomic requests are gone -- it is a latency chain per tile (point
// load, tap table, GEMM, gathers, barriers), not atomic-bound any more -- plus 0.10 ms for k_scatter_xy on rays along z: 1.08 -> 1.02 ms in
// total, but 0.89 -> 1.35 ms on oblique rays (k_scatter_xy's tap-by-tap path runs after this kernel instead of under it).  Deciding per tile
// HERE which form a tile takes (window test in the tap-table phase) cost this kernel 0.3 ms and lost everywhere.  (The same reduction INSIDE this kernel --
// window, M in LDS, 16 MFMAs per wavefront, between the gather phase and the sweep -- was built first and measured 1.09 -> 1.47 ms: the
// kernel is a latency chain per tile that lives on three blocks per CU, and the extra phase with its two barriers, or the spills it forces
// at 168 registers, costs more than the atomics it saves.)
template <int MODE, bool MM, int CT>
__global__ __launch_bounds__(256, MM ? (CT <= 96 ? 3 : 2) : 1) void k_voxel_sample_bwd(const GridParams g, const float* __restrict__ pts, long n,
                                                          const float* __restrict__ d_out, int d_stride, int d_col, GridGrads gg,
                                                          float* __restrict__ d_pts, const BinOut bo) {
    constexpr int STRD = CT + 1, FSTR = MM ? 33 : VSB_MAXF + 1;      // odd row strides (conflict-free column access)
    constexpr bool BINNED = MODE == 1, HYBRID = MODE == 2 || MODE == 3, XYDEF = MODE == 3;

    __shared__ float tfr[VS_SAMPLES * 3 * 6], dpt[VS_SAMPLES * 3];
    __shared__ int tax[VS_SAMPLES * 3 * 3];
    __shared__ __attribute__((aligned(16))) float pvs[VS_SAMPLES * STRD], lvs[VS_SAMPLES * STRD], dco[VS_SAMPLES * STRD],
        dout[VS_SAMPLES * FSTR], tw[VS_SAMPLES * VSB_TAPS];
    __shared__ int tix[VS_SAMPLES * VSB_TAPS];
    const int c0n = g.n_comp[0], c1n = g.n_comp[1], ctot = c0n + c1n + g.n_comp[2], F = g.app_dim, nbas = F * ctot;
    const int tid = threadIdx.x, ss = tid >> 7, ql = tid & 127;
    // this thread's channel in the (sample pair, 128 channel slots) sweeps of phase B: component group, channel inside it
    const int cg = ql < c0n ? 0 : (ql < c0n + c1n ? 1 : 2), cin = ql - (cg == 0 ? 0 : (cg == 1 ? c0n : c0n + c1n));
    const bool chan_on = ql < ctot;
    const bool rows16 = (c0n % 16 == 0) && (c1n % 16 == 0) && (g.n_comp[2] % 16 == 0);     // components = whole 16-lane DPP rows
    const float* gplane = sel3(cg, g.plane[0], g.plane[1], g.plane[2]);
    const float* gline = sel3(cg, g.line[0], g.line[1], g.line[2]);
    // ... and its (tap, channel) entries in the atomic sweeps: q = ql + 128 m over [4 plane taps x ctot | 2 line taps x ctot]
    constexpr int MQ = (6 * VS_MAXC + 127) / 128;
    int q_slot[MQ], q_c[MQ];
    float* q_ptr[MQ];
    bool q_plane[MQ];
#pragma unroll
    for (int m = 0; m < MQ; ++m) {
        const int q = ql + 128 * m;
        const bool on = q < 6 * ctot, pl = q < 4 * ctot;
        const int t = pl ? q / ctot : (q - 4 * ctot) / ctot, c = q % ctot;
        const int i = c < c0n ? 0 : (c < c0n + c1n ? 1 : 2), ci = c - (i == 0 ? 0 : (i == 1 ? c0n : c0n + c1n));
        q_plane[m] = pl;
        q_c[m] = c;
        q_slot[m] = pl ? 4 * i + t : 12 + 2 * i + t;
        float* base = pl ? sel3(i, gg.plane[0], gg.plane[1], gg.plane[2]) : sel3(i, gg.line[0], gg.line[1], gg.line[2]);
        q_ptr[m] = (on && base) ? base + ci : nullptr;
    }
    constexpr int NB = (VSB_MAXF * VS_MAXC + 255) / 256;
    float bacc[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) bacc[q] = 0.f;
    // The two small GEMMs of a tile (d coef = d out . basis and d basis += d out^T . coef: 32 x 32 x ctot each) run on the exact-float32
    // MFMA when app_dim is 32 and the channels come in 32-wide tiles: wavefront t owns channel tile t for both (on the VALU the second
    // one costs three LDS reads per multiply-add: 19 GB of LDS traffic per 2^19 samples).
    const int wv = tid >> 6, ln = tid & 63, mn = ln & 31, kb = ln >> 5;
    // MM (template: keeps the VALU fallback's 32 accumulators out of the common instantiation, which then fits two blocks per CU)
    const bool mm = MM, mm_wave = mm && wv * 32 < ctot;
    float bas_reg[16];
    f32x16 macc;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        macc[r] = 0.f;
        bas_reg[r] = mm_wave ? g.basis[(long)(2 * r + kb) * ctot + 32 * wv + mn] : 0.f;
    }
#ifdef EVD_SB_STAMP      // developer build: where does a tile's time go?  shader-clock cycles per phase, summed over the block's tiles (thread 0)
    long long sb_t[6] = {0, 0, 0, 0, 0, 0}, sb_last = __builtin_readcyclecounter();
#define SB_STAMP(k) { const long long now_ = __builtin_readcyclecounter(); sb_t[k] += now_ - sb_last; sb_last = now_; }
#else
#define SB_STAMP(k)
#endif
    for (long tile = blockIdx.x; tile * VS_SAMPLES < n; tile += gridDim.x) {
        const long s0 = tile * VS_SAMPLES;
        for (int o = tid; o < VS_SAMPLES * F; o += 256) {
            const int sl = o / F, f = o % F;
            dout[sl * FSTR + f] = s0 + sl < n ? d_out[(s0 + sl) * (long)d_stride + d_col + f] : 0.f;
        }
        if (tid < VS_SAMPLES * 3) {                // tap table: thread = (sample, component group)
            const int sl = tid / 3, i = tid % 3;
            const long s = s0 + sl < n ? s0 + sl : n - 1;
            const float pt[3] = {pts[s * 3], pts[s * 3 + 1], pts[s * 3 + 2]};
            VsItem it;
            const int grp0 = i == 0 ? 0 : (i == 1 ? c0n / 4 : (c0n + c1n) / 4);      // first 4-channel group of component i
            vs_issue<false>(g, pt, grp0, it);
            const bool live = s0 + sl < n;          // (the first group of a component has channel offset 0: ip / il address channel 0 of the tap)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                tix[sl * VSB_TAPS + 4 * i + t] = (int)it.ip[t];
                tw[sl * VSB_TAPS + 4 * i + t] = live ? it.wp[t] : 0.f;
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                tix[sl * VSB_TAPS + 12 + 2 * i + t] = (int)it.il[t];
                tw[sl * VSB_TAPS + 12 + 2 * i + t] = live ? it.wl[t] : 0.f;
            }
            float* fr = tfr + (sl * 3 + i) * 6;
            fr[0] = it.fw; fr[1] = it.fn; fr[2] = it.fl; fr[3] = it.kx; fr[4] = it.ky; fr[5] = it.kl;
            int* ta = tax + (sl * 3 + i) * 3;
            ta[0] = it.ax; ta[1] = it.ay; ta[2] = it.al;
            if constexpr (HYBRID) {
                if (live) {                         // the line taps of this (sample, component), for k_scatter_lines
                    const int C = sel3(i, g.n_comp[0], g.n_comp[1], g.n_comp[2]);
                    LTap lt_;
                    lt_.c0 = (int)it.il[0] / C; lt_.c1 = (int)it.il[1] / C; lt_.w0 = it.wl[0]; lt_.w1 = it.wl[1];
                    bo.ltap[s * 3 + i] = lt_;
                }
            }
            if constexpr (XYDEF) {
                if (i == 0) {                       // the x-y plane's taps of this sample, for k_scatter_xy (dead samples: zero weights)
                    const int Wp = g.grid[0];
                    const int cell0 = (int)(it.ip[0] / c0n), cell3 = (int)(it.ip[3] / c0n);
                    const int cy0 = cell0 / Wp, cx0 = cell0 - cy0 * Wp, cy1 = cell3 / Wp, cx1 = cell3 - cy1 * Wp;
                    PTap pt_;
                    pt_.cx0 = (unsigned short)cx0; pt_.cx1 = (unsigned short)cx1; pt_.cy0 = (unsigned short)cy0; pt_.cy1 = (unsigned short)cy1;
#pragma unroll
                    for (int t = 0; t < 4; ++t) pt_.w[t] = live ? it.wp[t] : 0.f;
                    if (live) bo.ptap[s] = pt_;
                }
            }
            if constexpr (BINNED) {
                if (live) {                         // tap records + the plane's tile key of this (sample, component)
                    const int C = sel3(i, g.n_comp[0], g.n_comp[1], g.n_comp[2]), Wp = sel3(i, g.grid[0], g.grid[0], g.grid[1]);
                    const int cell0 = (int)(it.ip[0] / C), cell3 = (int)(it.ip[3] / C);
                    const int cy0 = cell0 / Wp, cx0 = cell0 - cy0 * Wp, cy1 = cell3 / Wp, cx1 = cell3 - cy1 * Wp;
                    PTap pt_;
                    pt_.cx0 = (unsigned short)cx0; pt_.cx1 = (unsigned short)cx1; pt_.cy0 = (unsigned short)cy0; pt_.cy1 = (unsigned short)cy1;
#pragma unroll
                    for (int t = 0; t < 4; ++t) pt_.w[t] = it.wp[t];
                    bo.ptap[s * 3 + i] = pt_;
                    LTap lt_;
                    lt_.c0 = (int)(it.il[0] / C); lt_.c1 = (int)(it.il[1] / C); lt_.w0 = it.wl[0]; lt_.w1 = it.wl[1];
                    bo.ltap[s * 3 + i] = lt_;
                    unsigned* kp = sel3(i, bo.keys[0], bo.keys[1], bo.keys[2]);
                    kp[s] = (unsigned)((cy0 / SC_TS) * sel3(i, bo.tiles_x[0], bo.tiles_x[1], bo.tiles_x[2]) + cx0 / SC_TS);
                    if (i == 0) bo.ids[s] = (unsigned)s;
                }
            }
        }
        if (tid < VS_SAMPLES * 3) dpt[tid] = 0.f;
        __syncthreads();
        SB_STAMP(0);                                // d out rows, points, tap table
        if (mm) {
            if (mm_wave) {                          // D[sample][channel] = sum_f d out[sample][f] basis[f][channel]
                f32x16 a16;
#pragma unroll
                for (int r = 0; r < 16; ++r) a16[r] = 0.f;
#pragma unroll
                for (int j = 0; j < 16; ++j) a16 = __builtin_amdgcn_mfma_f32_32x32x2f32(dout[mn * FSTR + 2 * j + kb], bas_reg[j], a16, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) dco[((r & 3) + 8 * (r >> 2) + 4 * kb) * STRD + 32 * wv + mn] = a16[r];
            }
        } else {
            for (int o = tid; o < VS_SAMPLES * ctot; o += 256) {
                const int sl = o / ctot, c = o % ctot;
                float a = 0.f;
                for (int f = 0; f < F; ++f) a = fmaf(dout[sl * FSTR + f], g.basis[(long)f * ctot + c], a);
                dco[sl * STRD + c] = a;
            }
        }
        if (d_pts) __syncthreads();                 // the point gradient below reads d coef
        SB_STAMP(1);                                // d coef GEMM
        if (chan_on) {                              // pv, lv: lanes over channels, two samples per sweep
            // The taps of VSB_BATCH samples are loaded before the first is used.  (As one loop with "#pragma unroll 4" hipcc left it rolled --
            // the DPP row sums and LDS atomics of the d pts part are convergent operations --: six loads, then a wait for all six, 16 times
            // per tile, and in-kernel stamps put half of a tile's time in this phase.)
            for (int b0 = 0; b0 < VS_SAMPLES / 2; b0 += VSB_BATCH) {
            float Pb[VSB_BATCH][4], Lb[VSB_BATCH][2];
#pragma unroll
            for (int j = 0; j < VSB_BATCH; ++j) {
                const int* ti = tix + (ss + 2 * (b0 + j)) * VSB_TAPS;
#pragma unroll
                for (int t = 0; t < 4; ++t) Pb[j][t] = gplane[ti[4 * cg + t] + cin];
#pragma unroll
                for (int t = 0; t < 2; ++t) Lb[j][t] = gline[ti[12 + 2 * cg + t] + cin];
            }
#pragma unroll
            for (int j = 0; j < VSB_BATCH; ++j) {
                const int sl = ss + 2 * (b0 + j);
                const int* ti = tix + sl * VSB_TAPS;
                const float* w = tw + sl * VSB_TAPS;
                float pv = 0.f, lv = 0.f, P[4], Lt[2];
#pragma unroll
                for (int t = 0; t < 4; ++t) { P[t] = Pb[j][t]; pv = fmaf(w[4 * cg + t], P[t], pv); }
#pragma unroll
                for (int t = 0; t < 2; ++t) { Lt[t] = Lb[j][t]; lv = fmaf(w[12 + 2 * cg + t], Lt[t], lv); }
                pvs[sl * STRD + ql] = pv;
                lvs[sl * STRD + ql] = lv;
                if (d_pts) {
                    // d feature / d point through the interpolation weights (the ATen grid_sample backward: taps outside the grid
                    // contribute nothing), chained with d coef; summed over the channels of the wavefront, then over wavefronts in LDS
                    const float* fr = tfr + (sl * 3 + cg) * 6;
                    const float ww = fr[0], nn = fr[1], ee = 1.f - ww, sn = 1.f - nn;
#pragma unroll
                    for (int t = 0; t < 4; ++t) P[t] = w[4 * cg + t] != 0.f ? P[t] : 0.f;
                    const float dpx = (P[1] - P[0]) * sn + (P[3] - P[2]) * nn, dpy = (P[2] - P[0]) * ee + (P[3] - P[1]) * ww;
                    const float dl = (w[12 + 2 * cg + 1] != 0.f ? Lt[1] : 0.f) - (w[12 + 2 * cg] != 0.f ? Lt[0] : 0.f);
                    const float dc = dco[sl * STRD + ql];
                    float gx = dc * lv * dpx * fr[3], gy = dc * lv * dpy * fr[4], gl = dc * pv * dl * fr[5];
                    // sum over the component's channels.  When every component is a whole number of 16-lane rows (the shipped 64 / 16 / 16)
                    // the rows are summed in registers (DPP) and ONE lane per row adds to LDS: 18 LDS float atomics per sample instead of
                    // 288 -- ds_add_f32 runs at ~0.4 lane-operations per clock and CU on this chip (kernel_voxel_scatter.hip), so the
                    // 9216 of a tile cost more than everything else the tile does
                    const int* ta = tax + (sl * 3 + cg) * 3;
                    if (rows16) {
                        gx = row_sum_dpp(gx); gy = row_sum_dpp(gy); gl = row_sum_dpp(gl);
                        if ((tid & 15) == 0) {
                            atomicAdd(&dpt[sl * 3 + ta[0]], gx);
                            atomicAdd(&dpt[sl * 3 + ta[1]], gy);
                            atomicAdd(&dpt[sl * 3 + ta[2]], gl);
                        }
                    } else {
                        atomicAdd(&dpt[sl * 3 + ta[0]], gx);
                        atomicAdd(&dpt[sl * 3 + ta[1]], gy);
                        atomicAdd(&dpt[sl * 3 + ta[2]], gl);
                    }
                }
            }
            }
        }
        __syncthreads();
        SB_STAMP(2);                                // gathers, pv / lv, d pts
        // (Re-measured in round 2 with the half-tile walk that keeps a ray's runs together -- successive samples of an NDC ray address
        // ~12 distinct x-y cells and ~7 x / y line cells per 32 samples --: summing the run in a register before ONE atomic is 1.5-1.9x
        // SLOWER, 2.76 vs 1.46 ms at 2^19 samples: the walk is a chain of dependent LDS reads, the sweep below is not.)
        if constexpr (BINNED) {
            if (chan_on) {
                for (int sl = ss; sl < VS_SAMPLES && s0 + sl < n; sl += 2) {
                    const float dc = dco[sl * STRD + ql];
                    bo.rows_p[(s0 + sl) * ctot + ql] = dc * lvs[sl * STRD + ql];
                    bo.rows_l[(s0 + sl) * ctot + ql] = dc * pvs[sl * STRD + ql];
                }
            }
        } else {
            if constexpr (HYBRID) {
                if (chan_on) {
                    for (int sl = ss; sl < VS_SAMPLES && s0 + sl < n; sl += 2) bo.rows_l[(s0 + sl) * ctot + ql] = dco[sl * STRD + ql] * pvs[sl * STRD + ql];
                }
            }
            if constexpr (XYDEF) {
                if (ql < c0n) {
                    for (int sl = ss; sl < VS_SAMPLES && s0 + sl < n; sl += 2) bo.rows_p[(s0 + sl) * c0n + ql] = dco[sl * STRD + ql] * lvs[sl * STRD + ql];
                }
            }
            for (int sl = ss; sl < VS_SAMPLES; sl += 2) {
#pragma unroll
                for (int m = 0; m < MQ; ++m) {
                    if (!q_ptr[m] || (HYBRID && !q_plane[m]) || (XYDEF && q_slot[m] < 4)) continue;
                    const float w = tw[sl * VSB_TAPS + q_slot[m]];
                    if (w == 0.f) continue;
                    const int c = q_c[m];
                    const float other = q_plane[m] ? lvs[sl * STRD + c] : pvs[sl * STRD + c];
                    unsafeAtomicAdd(q_ptr[m] + tix[sl * VSB_TAPS + q_slot[m]], dco[sl * STRD + c] * other * w);
                }
            }
        }
        SB_STAMP(3);                                // rows + the atomic sweep (issue only: nothing waits for the adds here)
        if (d_pts && tid < VS_SAMPLES * 3 && s0 + tid / 3 < n) d_pts[(s0 + tid / 3) * 3 + tid % 3] = dpt[tid];
        if (gg.basis && mm) {
            if (mm_wave) {                          // D[f][channel] += sum_s d out[s][f] coef[s][channel]
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int sl = 2 * j + kb;
                    macc = __builtin_amdgcn_mfma_f32_32x32x2f32(dout[sl * FSTR + mn],
                                                                pvs[sl * STRD + 32 * wv + mn] * lvs[sl * STRD + 32 * wv + mn], macc, 0, 0, 0);
                }
            }
        } else if (gg.basis) {
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const int o = tid + 256 * q;
                if (o < nbas) {
                    const int f = o / ctot, c = o % ctot;
                    float a = bacc[q];
                    for (int sl = 0; sl < VS_SAMPLES; ++sl) a = fmaf(dout[sl * FSTR + f], pvs[sl * STRD + c] * lvs[sl * STRD + c], a);
                    bacc[q] = a;
                }
            }
        }
        __syncthreads();
        SB_STAMP(4);                                // basis_mat gradient GEMM, d pts rows, the tile's last barrier
    }
#ifdef EVD_SB_STAMP
    if constexpr (HYBRID) {
        if (tid == 0) {
            for (int k = 0; k < 5; ++k) bo.rows_l[blockIdx.x * 8 + k] = (float)sb_t[k];
            bo.rows_l[blockIdx.x * 8 + 5] = -7.f;
        }
    }
#endif
    if (gg.basis && mm) {
        if (mm_wave) {
#pragma unroll
            for (int r = 0; r < 16; ++r) unsafeAtomicAdd(gg.basis + (long)((r & 3) + 8 * (r >> 2) + 4 * kb) * ctot + 32 * wv + mn, macc[r]);
        }
    } else if (gg.basis) {
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int o = tid + 256 * q;
            if (o < nbas) unsafeAtomicAdd(gg.basis + o, bacc[q]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Round 3: the backward gather WAVEFRONT-AUTONOMOUS, like the forward's k_voxel_sample_w.  In-kernel stamps of the block-cooperative
// kernel above put half of a tile's time in its gather phase and showed the whole kernel to be a per-tile latency chain (point load, tap
// table, GEMM, gathers, four block barriers) that stays at ~0.72 ms per 2^19 samples even with two thirds of its atomics removed.  Here a
// WAVEFRONT owns 16 consecutive samples (of one ray, as the renderer lays them out) from the point load to its last atomic; the one
// block barrier orders the basis_mat image in LDS:
//   0  tap geometry of its 48 (sample, component) pairs on 48 lanes -> its LDS slice (+ the line tap records for k_scatter_lines)
//   1  d coef^T = basis^T . d out^T on v_mfma_f32_16x16x4_f32 (6 channel tiles x 8 steps; d out as the register-resident B operand)
//   2  the gather exactly as the forward does it: 3 items (sample, 8 channels) per lane, 36 16-byte loads in flight, then per item
//      pv, lv;  line rows d coef pv -> HBM (k_scatter_lines);  coef = pv lv -> HBM (k_basis_grad);  plane rows d coef lv -> LDS in place;
//      the point gradient's per-item partial sums -> LDS
//   3  the plane taps, lanes over channels (every atomic instruction covers whole 64-byte runs).  A 64-channel component (the x-y
//      plane) is walked sample by sample with the sum kept in a REGISTER while successive samples address the same cell -- the rays of an
//      NDC scene run along z, a tile's 16 samples touch 1-4 x-y cells -- and flushed by one atomic per (run, tap); the 16 / 32-channel
//      components (their taps move with every sample) add tap by tap
//   4  the point gradient: 48 lanes sum the partials of their (sample, axis)
// The basis_mat gradient d out^T . coef is a plain GEMM over all samples and runs as its own small kernel (k_basis_grad) on the coefficient
// rows written in phase 2; the line taps go through k_scatter_lines as in the hybrid form.
constexpr int VBW_SAMPLES = 16, VBW_WAVES = 4;
constexpr int VBW_BSTR = 112;                   // basis_mat row stride in LDS: 16 (mod 32) words, so that the MFMA A reads (lane = channel + 16 x row step) hit 64 banks
constexpr int VBW_CSTR = 97;                    // d coef / plane-row stride (ctot <= 96), odd: lanes over channels read conflict-free
constexpr int VBW_MAXG = 12;                    // 8-channel groups per sample
struct VbwTaps {
    int ip[4], il[2];                           // element offsets of channel 0 of the taps (clamped)
    float wp[4], wl[2];                         // interpolation weights, 0 = outside (zero padding) or dead sample
    float fw, fn, kx, ky, kl;                   // fractional position in the plane cell; d (pixel coordinate) / d (point coordinate)
    int pad;
};
constexpr int VBW_LROW = 33;                    // line rows of components 1 and 2 kept in LDS ([16][33]: n_comp[1] + n_comp[2] <= 32)
// a wavefront's slice: tap tables | d coef -> plane rows [16][CSTR] | point-gradient partial sums [16][3 quads][3] | the coefficient rows
// pv lv [16][96] of the in-kernel basis gradient (the LINES12 form keeps its line rows [16][LROW] there and the coefficients in registers).
// 16 KiB per wavefront: two workgroups of four per CU (2 x 78 KiB of the 160 KiB)
constexpr int VBW_FSTR = 96;                    // coefficient row stride: the MFMA B reads (32 channels x 2 samples per step) cover the 64 banks
constexpr size_t VBW_SLICE = VBW_SAMPLES * 3 * sizeof(VbwTaps) + (size_t)VBW_SAMPLES * VBW_CSTR * 4 + (size_t)VBW_SAMPLES * 9 * 4 + (size_t)VBW_SAMPLES * VBW_FSTR * 4;
static_assert((size_t)VBW_SAMPLES * VBW_LROW * 4 <= (size_t)VBW_SAMPLES * VBW_FSTR * 4 && VBW_MAXG * 8 <= VBW_FSTR, "line rows alias the coefficient rows");
constexpr size_t VBW_LDS = (size_t)32 * VBW_BSTR * 4 + VBW_WAVES * VBW_SLICE;
// ISS (round 6): the last wavefront of the workgroup ISSUES the plane taps' atomics of the other three.  The VM counter retires in order, so a
// wavefront that adds its own taps cannot start the next tile's loads before its ~40 atomic instructions have retired: with everything else
// removed the kernel's atomics take 0.20 ms per 2^19 samples, everything but the atomics 0.2x ms, together 0.42 ms (profiles/r06_scatter_ablation.log)
// -- the two do not overlap inside a wavefront.  A compute wavefront now leaves a tile's plane rows and tap cells in a hand-off buffer in LDS
// ([16][96] rows | cells | weights: 7.5 KiB) and goes on; the issuer walks the buffers of its three producers (run-length merge as before) and
// is the only wavefront whose VM counter carries atomics.  The coefficients pv lv of the basis gradient replace d coef in place (no separate
// rows): the slice shrinks to 10 KiB, 66.5 KiB per workgroup, two workgroups per CU as before.
constexpr int VBI_CW = VBW_WAVES - 1;                                                                // compute wavefronts of an issuer-form workgroup
constexpr int VBI_HROW = 96;                                                                          // hand-off row stride (ctot <= 96)
constexpr size_t VBI_SLICE = VBW_SAMPLES * 3 * sizeof(VbwTaps) + (size_t)VBW_SAMPLES * VBW_CSTR * 4 + (size_t)VBW_SAMPLES * 9 * 4 + 64;
constexpr size_t VBI_HAND = (size_t)VBW_SAMPLES * VBI_HROW * 4 + (size_t)VBW_SAMPLES * 3 * 4 * 8;   // rows | int cell[48][4] | float weight[48][4]
constexpr size_t VBI_LDS = (size_t)32 * VBW_BSTR * 4 + VBI_CW * (VBI_SLICE + VBI_HAND) + 64;
static_assert(VBI_SLICE % 16 == 0 && VBI_HAND % 16 == 0 && 2 * VBI_LDS <= 160 * 1024, "two issuer-form workgroups per CU");
static_assert(sizeof(VbwTaps) % 8 == 0 && VBW_SLICE % 16 == 0, "slice alignment");

// vs_geometry + the quantities the point gradient needs; same formulas, same order (the forward's weights bit for bit)
__device__ __forceinline__ void vbw_geometry(const GridParams& g, const float (&pt)[3], int i, bool live, VbwTaps& tp) {
    float xyz[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) xyz[c] = __fsub_rn(__fmul_rn(__fsub_rn(pt[c], g.aabb_min[c]), g.inv[c]), 1.f);   // voxnerf.py:205
    const int C = sel3(i, g.n_comp[0], g.n_comp[1], g.n_comp[2]);
    const int Wp = sel3(i, g.grid[0], g.grid[0], g.grid[1]);
    const int Hp = sel3(i, g.grid[1], g.grid[2], g.grid[2]);
    const int Lp = sel3(i, g.grid[2], g.grid[1], g.grid[0]);
    const float cx = sel3(i, xyz[0], xyz[0], xyz[1]), cy = sel3(i, xyz[1], xyz[2], xyz[2]), cl = sel3(i, xyz[2], xyz[1], xyz[0]);
    const float ix = unnorm(cx, Wp), iy = unnorm(cy, Hp);
    const float fx = fminf(fmaxf(floorf(ix), -2.f), (float)Wp), fy = fminf(fmaxf(floorf(iy), -2.f), (float)Hp);
    const float ww = __fsub_rn(ix, floorf(ix)), ee = __fsub_rn(1.f, ww), nn = __fsub_rn(iy, floorf(iy)), ss = __fsub_rn(1.f, nn);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const bool vx0 = x0 >= 0 && x0 < Wp, vx1 = x1 >= 0 && x1 < Wp, vy0 = y0 >= 0 && y0 < Hp, vy1 = y1 >= 0 && y1 < Hp;
    const int cx0 = min(max(x0, 0), Wp - 1), cx1 = min(max(x1, 0), Wp - 1), cy0 = min(max(y0, 0), Hp - 1), cy1 = min(max(y1, 0), Hp - 1);
    tp.ip[0] = (cy0 * Wp + cx0) * C;
    tp.ip[1] = (cy0 * Wp + cx1) * C;
    tp.ip[2] = (cy1 * Wp + cx0) * C;
    tp.ip[3] = (cy1 * Wp + cx1) * C;
    tp.wp[0] = (live && vy0 && vx0) ? __fmul_rn(ee, ss) : 0.f;
    tp.wp[1] = (live && vy0 && vx1) ? __fmul_rn(ww, ss) : 0.f;
    tp.wp[2] = (live && vy1 && vx0) ? __fmul_rn(ee, nn) : 0.f;
    tp.wp[3] = (live && vy1 && vx1) ? __fmul_rn(ww, nn) : 0.f;
    const float il = unnorm(cl, Lp);
    const float fl = fminf(fmaxf(floorf(il), -2.f), (float)Lp);
    const float ln = __fsub_rn(il, floorf(il)), ls = __fsub_rn(1.f, ln);
    const int l0 = (int)fl, l1 = l0 + 1;
    tp.il[0] = min(max(l0, 0), Lp - 1) * C;
    tp.il[1] = min(max(l1, 0), Lp - 1) * C;
    tp.wl[0] = (live && l0 >= 0 && l0 < Lp) ? ls : 0.f;
    tp.wl[1] = (live && l1 >= 0 && l1 < Lp) ? ln : 0.f;
    tp.fw = ww; tp.fn = nn;
    tp.kx = 0.5f * (float)(Wp - 1) * sel3(i, g.inv[0], g.inv[0], g.inv[1]);
    tp.ky = 0.5f * (float)(Hp - 1) * sel3(i, g.inv[1], g.inv[2], g.inv[2]);
    tp.kl = 0.5f * (float)(Lp - 1) * sel3(i, g.inv[2], g.inv[1], g.inv[0]);
    tp.pad = 0;
}

// LINES12: the line taps of components 1 and 2 (the x / y lines of an NDC scene: one cell for a whole run of samples) are summed along
// runs and added here, like the plane taps; only component 0's line (the z line, a new cell every sample) leaves as rows for k_scatter_lines
// BAS (round 4): the basis_mat gradient d out^T . coef INSIDE this kernel.  The workgroups are persistent (a wavefront walks tiles
// blockIdx.x, blockIdx.x + gridDim.x, ...: basis_mat is staged in LDS once per workgroup instead of once per 64 samples), a wavefront keeps
// the 8 coefficients of each of its three gather items in registers over the plane-tap phase, puts them where the plane rows were
// (its slice's d coef array is free by then) and adds its 16 samples' [F x ctot] product to 3 x 16 accumulator registers on
// v_mfma_f32_32x32x2_f32 (d out rows as the A operand straight from L2); one fold through LDS + one atomic flush per workgroup at the end.
// Gone: the coefficient rows [n, ctot] (201 MB written and read back per 2^19 samples) and the k_basis_grad launch.
// The plane-tap walk (phase 3 of k_voxel_sample_bwd_w): a lane owns a (tap, channel), walks the tile's 16 samples with the sum of a RUN of
// samples on one cell in a register and adds it once per run.  Round 6 -- the kernel is bound by the number of instructions its two wavefronts per
// SIMD issue (~5.5 k per tile and wavefront, 4 cycles each; a build WITHOUT the atomics showed the walk alone at 33.6 k of a tile's 51.8 k
// cycles, profiles/r06_scatter_stamps_before_walk_rewrite.log: the "atomic phase" was this loop, not the atomics), so the walk is cut to what
// it needs:
//   * all LDS operands of a pass are fetched first (independent reads), the walk runs on registers;
//   * weight x row with the legacy multiply (0 x anything = 0): the same sums as the guarded form `w != 0 ? w * r : 0` of rounds 3-5 -- a tap
//     outside the grid (weight 0) adds nothing even where the row is not finite -- without a compare and a select per step;
//   * a run whose sum is exactly 0 in a lane adds nothing (x + 0 = x): the flag `any sample live` of rounds 3-5 is that test.
// vbw_walk_pass: one tap per lane group (a 16-channel plane: all four taps in one pass of 64 lanes).
template <class FW, class FC, class FR>
__device__ __forceinline__ void vbw_walk_pass(float* __restrict__ gp, int c, bool act, FW fw, FC fc, FR fr) {
    float w[VBW_SAMPLES], r[VBW_SAMPLES];
    int cell[VBW_SAMPLES];
#pragma unroll
    for (int sm = 0; sm < VBW_SAMPLES; ++sm) { w[sm] = fw(sm); cell[sm] = fc(sm); r[sm] = fr(sm); }
    float acc = 0.f;
#pragma unroll
    for (int sm = 0; sm < VBW_SAMPLES; ++sm) {
        acc += vbw_mul_legacy(w[sm], r[sm]);
        const bool flush = sm == VBW_SAMPLES - 1 || cell[sm + 1 < VBW_SAMPLES ? sm + 1 : sm] != cell[sm];
        if (flush) {
#ifdef EVD_VBW_NO_ATOMICS     // developer ablation: everything but the plane taps' atomics (the sum is kept alive)
            if (act && acc == 1.2345e38f) gp[cell[sm] + c] = acc;
#else
            if (act && acc != 0.f) unsafeAtomicAdd(gp + cell[sm] + c, acc);
#endif
            acc = 0.f;
        }
    }
}

// vbw_walk_plane64: the 64-channel plane, lane = channel, ALL FOUR taps in one pass.  The taps of a sample are the corners of one cell, so
// the four cell indices change together: the run ends are the steps where tap 0's or tap 3's index changes (both unchanged <=> the
// clamped corner pairs (x0, y0) and (x1, y1) unchanged <=> all four unchanged), decided on two scalar registers per sample.  Per step: four
// multiplies and four adds; the row value is read once instead of four times.
template <class FW4, class FC, class FR>
__device__ __forceinline__ void vbw_walk_plane64(float* __restrict__ gp, int c, FW4 fw4, FC fc, FR fr) {
    float r[VBW_SAMPLES];
    f32x4 w[VBW_SAMPLES];
    int c0[VBW_SAMPLES], c3[VBW_SAMPLES];
#pragma unroll
    for (int sm = 0; sm < VBW_SAMPLES; ++sm) { w[sm] = fw4(sm); c0[sm] = fc(sm, 0); c3[sm] = fc(sm, 3); r[sm] = fr(sm); }
#pragma unroll
    for (int sm = 0; sm < VBW_SAMPLES; ++sm) { c0[sm] = __builtin_amdgcn_readfirstlane(c0[sm]); c3[sm] = __builtin_amdgcn_readfirstlane(c3[sm]); }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sm = 0; sm < VBW_SAMPLES; ++sm) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] += vbw_mul_legacy(w[sm][t], r[sm]);
        const int nx = sm + 1 < VBW_SAMPLES ? sm + 1 : sm;
        const bool flush = sm == VBW_SAMPLES - 1 || c0[nx] != c0[sm] || c3[nx] != c3[sm];
        if (flush) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int cell = fc(sm, t);
#ifdef EVD_VBW_NO_ATOMICS
                if (acc[t] == 1.2345e38f) gp[cell + c] = acc[t];
#else
                if (acc[t] != 0.f) unsafeAtomicAdd(gp + cell + c, acc[t]);
#endif
                acc[t] = 0.f;
            }
        }
    }
}

// HALF (round 6): phase 2 re-gathers the grid values from the FLOAT16 copies (GridParams::plane_h / line_h) -- the values the forward of the
// half-precision arithmetic modes interpolated (evd_voxel_api.hip grids_half_for), so the products d coef x value are the gradient of the function
// that forward computed; half the gather's loads and bytes, weight x value + sum as one v_fma_mix_f32 on the float16 value (as k_voxel_sample_m).
template <bool DPTS, bool LINES12, bool BAS, bool ISS = false, bool HALF = false>
__global__ __launch_bounds__(64 * VBW_WAVES, 2) void k_voxel_sample_bwd_w(const GridParams g, const float* __restrict__ pts, long n,
                                                                          const float* __restrict__ d_out, int d_stride, int d_col, GridGrads gg,
                                                                          float* __restrict__ d_pts, float* __restrict__ rows_l, LTap* __restrict__ ltap,
                                                                          float* __restrict__ coef_out, unsigned* __restrict__ lmax) {
    extern __shared__ __attribute__((aligned(16))) char vbw_smem[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c0n = g.n_comp[0], c1n = g.n_comp[1], c2n = g.n_comp[2], ctot = c0n + c1n + c2n, F = g.app_dim;
    static_assert(!ISS || (BAS && !LINES12), "the issuer form is the persistent kernel with the basis gradient inside, lines through k_scatter_lines");
    constexpr int NCW = ISS ? VBI_CW : VBW_WAVES;                                    // wavefronts that own tiles
    float* bs = reinterpret_cast<float*>(vbw_smem);                                  // basis_mat [32][VBW_BSTR], rows >= F are zero
    char* slice = vbw_smem + (size_t)32 * VBW_BSTR * 4 + (size_t)(wv < NCW ? wv : 0) * (ISS ? VBI_SLICE : VBW_SLICE);
    VbwTaps* taps = reinterpret_cast<VbwTaps*>(slice);
    float* dco = reinterpret_cast<float*>(slice + VBW_SAMPLES * 3 * sizeof(VbwTaps));   // d coef [16][VBW_CSTR], later the plane rows d coef lv
    float* dpart = dco + VBW_SAMPLES * VBW_CSTR;                                     // [16][3 quads of 8-channel groups][3 axes] d pts partial sums
    float* cfl = ISS ? dco : dpart + VBW_SAMPLES * 9;                                // [16][VBW_FSTR] coefficients pv lv (BAS); ISS: in place of d coef
    float* lrow = cfl;                                                               // [16][VBW_LROW] line rows d coef pv of components 1, 2 (LINES12)
    constexpr bool CF_LDS = BAS && !LINES12;
    constexpr int CFSTR = ISS ? VBW_CSTR : VBW_FSTR;                                 // row stride of the coefficient rows
    // ISS: hand-off buffers [NCW] behind the slices, then the flags (0 free, 1 full, 2 producer finished)
    char* hand0 = vbw_smem + (size_t)32 * VBW_BSTR * 4 + (size_t)VBI_CW * VBI_SLICE;
    float* hrow = reinterpret_cast<float*>(hand0 + (size_t)(wv < NCW ? wv : 0) * VBI_HAND);
    int* hcell = reinterpret_cast<int*>(hrow + VBW_SAMPLES * VBI_HROW);
    float* hwgt = reinterpret_cast<float*>(hcell + VBW_SAMPLES * 3 * 4);
    volatile int* flags = reinterpret_cast<volatile int*>(hand0 + (size_t)VBI_CW * VBI_HAND);
    if (ISS && threadIdx.x < 16) flags[threadIdx.x] = 0;
    const int ng = ctot / 8;
    // basis_mat -> LDS (the block's only shared state) as the A operands of phase 1 (round 6): d coef^T = basis^T . d out^T on
    // v_mfma_f32_16x16x32_f16 in the split form (hi = f16(x), lo = f16(x - hi): A_hi B_hi + A_hi B_lo + A_lo B_hi, 2^-21 per product) --
    // 18 MFMAs of 16 cycles per tile instead of 48 float32 16 x 16 x 4 of 32.  Entry (channel tile ct, lane): basis[8 (lane / 16) + j][16 ct + lane % 16],
    // j = 0 .. 7; the region is the one the block's fold of the basis gradient uses at the end (bs).  float16 has 5 exponent bits and gradients
    // are small, so both operands are brought to [2^13, 2^14) by a power of two first -- one per channel (row of A, kept in a1_inv) and
    // one per sample (column of B, vbw_pow2_scale on the row's largest magnitude) -- and the product is scaled back exactly.
    vm_h8* a1_hi = reinterpret_cast<vm_h8*>(bs);                                     // [6][64]
    vm_h8* a1_lo = a1_hi + 6 * 64;
    float* a1_inv = reinterpret_cast<float*>(a1_lo + 6 * 64);                        // [96]
    static_assert((size_t)2 * 6 * 64 * 16 + 96 * 4 <= (size_t)32 * VBW_BSTR * 4, "the split operands fit the fold buffer");
    for (int e = threadIdx.x; e < 6 * 64; e += 64 * VBW_WAVES) {                     // whole wavefronts: the shuffles below see all four k groups of a channel
        const int l = e & 63, ct = e >> 6, ch = 16 * ct + (l & 15), f0 = 8 * (l >> 4);
        float v[8], m = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[j] = (f0 + j < F && ch < ctot) ? g.basis[(long)(f0 + j) * ctot + ch] : 0.f;
            m = fmaxf(m, fabsf(v[j]));
        }
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        float inv;
        const float sc = vbw_pow2_scale(m, &inv);
        vm_h8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = v[j] * sc;
            hi[j] = (_Float16)x;
            lo[j] = (_Float16)(x - (float)hi[j]);
        }
        a1_hi[e] = hi;
        a1_lo[e] = lo;
        if (l < 16) a1_inv[ch] = inv;
    }
    __syncthreads();                              // the only block-wide barrier in front of the tiles: basis_mat visible
    auto wave_sync = []() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    constexpr int NCT = 3;                        // 32-channel tiles of the basis gradient (ctot <= 96)
    f32x16 bacc[NCT];
    if (BAS) {
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) bacc[c][r] = 0.f;
    }
#ifdef EVD_VBW_STAMP    // developer build (tools/dev/stamp_scatter_w.py): shader-clock cycles of this wavefront's phases, summed over its tiles
    long long tph[7] = {0, 0, 0, 0, 0, 0, 0}, tq0, tq1;
#define EVD_VBW_T0() tq0 = __builtin_readcyclecounter()
#define EVD_VBW_T(i) { tq1 = __builtin_readcyclecounter(); tph[i] += tq1 - tq0; tq0 = tq1; }
#else
#define EVD_VBW_T0()
#define EVD_VBW_T(i)
#endif
    float rmaxv = 0.f;                            // max |line row value| this lane wrote (k_scatter_lines' fixed-point scale: saves it a pass over the rows)
    const long wtiles = (n + VBW_SAMPLES - 1) / VBW_SAMPLES;
    auto lds_done = []() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
    if (ISS && wv >= NCW) {
        // the issuer: polls its producers' flags; a full buffer's taps are walked exactly like phase 3 below (lanes = (tap, channel), the sum
        // of a run of samples on one cell in a register, one atomic per run), then the buffer is handed back.  No load of this wavefront ever
        // waits behind its atomics: it has none but LDS reads.
        int fin = 0;
        while (fin != (1 << NCW) - 1) {
            bool idle = true;
#pragma unroll 1
            for (int w = 0; w < NCW; ++w) {
                if (fin >> w & 1) continue;
                const int f = __builtin_amdgcn_readfirstlane(flags[w]);
                if (f == 2) { fin |= 1 << w; continue; }
                if (f != 1) continue;
                asm volatile("" ::: "memory");      // (no read of the buffer may be moved in front of the flag's)
                idle = false;
                const float* hr = reinterpret_cast<const float*>(hand0 + (size_t)w * VBI_HAND);
                const int* hc = reinterpret_cast<const int*>(hr + VBW_SAMPLES * VBI_HROW);
                const float* hw = reinterpret_cast<const float*>(hc + VBW_SAMPLES * 3 * 4);
                int coff = 0;
#pragma unroll 1
                for (int i = 0; i < 3; ++i) {
                    const int C = sel3(i, c0n, c1n, c2n);
                    float* gp = sel3(i, gg.plane[0], gg.plane[1], gg.plane[2]);
                    if (gp) {
                        const int tpp = 64 / C < 4 ? 64 / C : 4, j = lane / C, c = lane % C;
                        if (C == 64) {
                            vbw_walk_plane64(gp, lane, [&](int sm) { return *reinterpret_cast<const f32x4*>(hw + (sm * 3 + i) * 4); },
                                             [&](int sm, int t) { return hc[(sm * 3 + i) * 4 + t]; }, [&](int sm) { return hr[sm * VBI_HROW + coff + lane]; });
                        } else {
#pragma unroll 1
                            for (int t0 = 0; t0 < 4; t0 += tpp) {
                                const int t = t0 + (j < tpp ? j : 0);
                                vbw_walk_pass(gp, c, j < tpp, [&](int sm) { return hw[(sm * 3 + i) * 4 + t]; }, [&](int sm) { return hc[(sm * 3 + i) * 4 + t]; },
                                              [&](int sm) { return hr[sm * VBI_HROW + coff + c]; });
                            }
                        }
                    }
                    coff += C;
                }
                lds_done();                           // every read of the buffer has returned
                if (lane == 0) flags[w] = 0;
            }
            if (idle) __builtin_amdgcn_s_sleep(8);
        }
    }
    for (long wt = (long)blockIdx.x * NCW + wv; wv < NCW && wt < wtiles; wt += (long)gridDim.x * NCW) {
    const long s0 = wt * VBW_SAMPLES;
    EVD_VBW_T0();
    // d out as the MFMA B operand: lane (col = sample, kh) holds d out[sample][4 step + kh]
    const int col = lane & 15, kh = lane >> 4;
    float dv[8];                                  // lane (col = sample, kh): d out[sample][8 kh .. 8 kh + 7]
    {
        const long s = s0 + col;
        const float* r = d_out + (s < n ? s : n - 1) * (long)d_stride + d_col;
#pragma unroll
        for (int j = 0; j < 8; ++j) dv[j] = (s < n && 8 * kh + j < F) ? r[8 * kh + j] : 0.f;
    }
    if (lane < VBW_SAMPLES * 3) {                 // phase 0: geometry of this wavefront's (sample, component) pairs
        const int sl = lane / 3, i = lane % 3;
        const bool live = s0 + sl < n;
        const long s = live ? s0 + sl : n - 1;
        const float pt[3] = {pts[s * 3], pts[s * 3 + 1], pts[s * 3 + 2]};
        VbwTaps tp;
        vbw_geometry(g, pt, i, live, tp);
        taps[lane] = tp;
        if (live && ltap && !(LINES12 && i > 0)) {
            const int C = sel3(i, c0n, c1n, c2n);
            LTap lt_;
            lt_.c0 = tp.il[0] / C; lt_.c1 = tp.il[1] / C; lt_.w0 = tp.wl[0]; lt_.w1 = tp.wl[1];
            ltap[s * 3 + i] = lt_;
        }
    }
    wave_sync();                                  // the tap tables are the wavefront's own
    EVD_VBW_T(0);
#ifdef EVD_VBW_ATOMICS_ONLY      // developer ablation (tools/dev/scatter_atomics_only.sh): geometry + the plane-tap phase on constant rows -- what the atomics of the REAL address stream cost alone
    for (int o = lane; o < VBW_SAMPLES * VBW_CSTR; o += 64) dco[o] = 1.f;
    f32x4 cfk[1][2];
    constexpr int UNR = 1, TRIPS = 1;
    const int items = 0;
#else
    // phase 1: D[channel 16 ct + 4 kh + r][sample col] = sum_f basis[f][channel] d out[sample][f]
    {
        float m = 0.f, binv;
#pragma unroll
        for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(dv[j]));
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        const float bsc = vbw_pow2_scale(m, &binv);
        vm_h8 bh, bl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = dv[j] * bsc;
            bh[j] = (_Float16)x;
            bl[j] = (_Float16)(x - (float)bh[j]);
        }
        for (int ct = 0; ct < ctot / 16; ++ct) {
            const vm_h8 ah = a1_hi[ct * 64 + lane], al = a1_lo[ct * 64 + lane];
            const f32x4 ai = *reinterpret_cast<const f32x4*>(a1_inv + 16 * ct + 4 * kh);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = acc[r] * ai[r] * binv;
#pragma unroll
            for (int r = 0; r < 4; ++r) dco[col * VBW_CSTR + 16 * ct + 4 * kh + r] = acc[r];
        }
    }
    wave_sync();
    EVD_VBW_T(1);
    // phase 2: gather, 3 items per lane in flight
    const int items = VBW_SAMPLES * ng;
    // items in flight per lane and trip: three (144 registers of raw taps) -- two where the wavefront also carries the basis accumulators
    // (48 registers) AND the point gradient's operands: at three that form spills 27 registers into the tile loop
    constexpr int UNR = (BAS && DPTS && !HALF) ? 2 : 3, TRIPS = 3 / UNR + (3 % UNR ? 1 : 0);      // (HALF: 24 instead of 48 registers of raw taps per item)
    f32x4 cfk[TRIPS * UNR][2];                    // BAS: the coefficients of this lane's items (ng <= 12: items <= 3 x 64)
#pragma unroll
    for (int trip = 0; trip < TRIPS; ++trip) {
        const int base = lane + trip * UNR * 64;
        if (base >= items) break;
        f32x4 rawp[UNR][4][2], rawl[UNR][2][2];
        vm_h8 hfp[UNR][4], hfl[UNR][2];           // HALF: the taps' eight float16 values (one 16-byte load each)
        int sl[UNR], grp[UNR], comp[UNR];
        bool on[UNR];
#pragma unroll
        for (int q = 0; q < UNR; ++q) {
            const int t = base + q * 64;
            on[q] = t < items;
            sl[q] = on[q] ? t / ng : 0;
            grp[q] = on[q] ? t % ng : 0;
            int i = 0, c8 = grp[q] * 8;
            if (c8 >= c0n) { c8 -= c0n; i = 1; if (c8 >= c1n) { c8 -= c1n; i = 2; } }
            comp[q] = i;
            const VbwTaps& tp = taps[sl[q] * 3 + i];
            if (HALF) {
                const _Float16* plh = sel3(i, g.plane_h[0], g.plane_h[1], g.plane_h[2]) + c8;
                const _Float16* lih = sel3(i, g.line_h[0], g.line_h[1], g.line_h[2]) + c8;
#pragma unroll
                for (int k = 0; k < 4; ++k) hfp[q][k] = *reinterpret_cast<const vm_h8*>(plh + tp.ip[k]);
#pragma unroll
                for (int k = 0; k < 2; ++k) hfl[q][k] = *reinterpret_cast<const vm_h8*>(lih + tp.il[k]);
            } else {
                const float* pl = sel3(i, g.plane[0], g.plane[1], g.plane[2]) + c8;
                const float* li = sel3(i, g.line[0], g.line[1], g.line[2]) + c8;
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int v = 0; v < 2; ++v) rawp[q][k][v] = *reinterpret_cast<const f32x4*>(pl + tp.ip[k] + 4 * v);
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int v = 0; v < 2; ++v) rawl[q][k][v] = *reinterpret_cast<const f32x4*>(li + tp.il[k] + 4 * v);
            }
        }
        if (ISS && trip == 0) {                   // (the gather's loads are in flight) the issuer is done with the previous tile's buffer
            while (__builtin_amdgcn_readfirstlane(flags[wv]) != 0) __builtin_amdgcn_s_sleep(2);
        }
#pragma unroll
        for (int q = 0; q < UNR; ++q) {
            const VbwTaps& tp = taps[sl[q] * 3 + comp[q]];
            const bool live = on[q] && s0 + sl[q] < n;
            const int cb = grp[q] * 8;
            float* drow = dco + sl[q] * VBW_CSTR + cb;
            float gx = 0.f, gy = 0.f, gl = 0.f;
            const float ww = tp.fw, nn = tp.fn, ee = 1.f - ww, sn = 1.f - nn;
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                f32x4 pv = {0.f, 0.f, 0.f, 0.f}, lv = {0.f, 0.f, 0.f, 0.f};
                // (round 6: w != 0 ? pv + raw w : pv as pv + legacy(w, raw) -- the same value, a tap outside the grid (w = 0) adds 0 whatever
                // lies at its clamped address -- one instruction less per element in a kernel bound by the instructions it issues)
                auto plane_val = [&](int t, int k) __attribute__((always_inline)) { return HALF ? (float)hfp[q][t][4 * v + k] : rawp[q][t][v][k]; };
                auto line_val = [&](int t, int k) __attribute__((always_inline)) { return HALF ? (float)hfl[q][t][4 * v + k] : rawl[q][t][v][k]; };
                if (HALF && DPTS) {               // the converted values are needed for the point gradient anyway: conversions + packed FMAs
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int k = 0; k < 4; ++k) pv[k] = __builtin_fmaf(tp.wp[t], plane_val(t, k), pv[k]);
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int k = 0; k < 4; ++k) lv[k] = __builtin_fmaf(tp.wl[t], line_val(t, k), lv[k]);
                } else if (HALF) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const vbw_u32x4 pw = __builtin_bit_cast(vbw_u32x4, hfp[q][t]);
#pragma unroll
                        for (int k = 0; k < 4; k += 2) {
                            pv[k] = vbw_fma_mix<0>(tp.wp[t], pw[2 * v + (k >> 1)], pv[k]);
                            pv[k + 1] = vbw_fma_mix<1>(tp.wp[t], pw[2 * v + (k >> 1)], pv[k + 1]);
                        }
                    }
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const vbw_u32x4 lw = __builtin_bit_cast(vbw_u32x4, hfl[q][t]);
#pragma unroll
                        for (int k = 0; k < 4; k += 2) {
                            lv[k] = vbw_fma_mix<0>(tp.wl[t], lw[2 * v + (k >> 1)], lv[k]);
                            lv[k + 1] = vbw_fma_mix<1>(tp.wl[t], lw[2 * v + (k >> 1)], lv[k + 1]);
                        }
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int k = 0; k < 4; ++k) pv[k] = __fadd_rn(pv[k], vbw_mul_legacy(tp.wp[t], plane_val(t, k)));
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int k = 0; k < 4; ++k) lv[k] = __fadd_rn(lv[k], vbw_mul_legacy(tp.wl[t], line_val(t, k)));
                }
                f32x4 dc, rl, cf, rp;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    dc[k] = drow[4 * v + k];
                    rl[k] = dc[k] * pv[k];
                    cf[k] = pv[k] * lv[k];
                    rp[k] = dc[k] * lv[k];
                }
                if (LINES12 && comp[q] > 0) {
                    if (on[q]) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) lrow[sl[q] * VBW_LROW + cb - c0n + 4 * v + k] = rl[k];
                    }
                } else if (live && rows_l) {
                    *reinterpret_cast<f32x4*>(rows_l + (s0 + sl[q]) * ctot + cb + 4 * v) = rl;
                    if (lmax) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float a = fabsf(rl[k]);
                            rmaxv = a != a ? __builtin_huge_valf() : fmaxf(rmaxv, a);       // (a NaN is recorded as +inf)
                        }
                    }
                }
                if (ISS) {                        // the plane rows into the hand-off buffer, the coefficients in place of d coef
                    if (on[q]) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            hrow[sl[q] * VBI_HROW + cb + 4 * v + k] = rp[k];
                            drow[4 * v + k] = cf[k];
                        }
                    }
                } else {
                if (CF_LDS) {
                    if (on[q]) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) cfl[sl[q] * VBW_FSTR + cb + 4 * v + k] = cf[k];
                    }
                } else if (BAS) cfk[trip * UNR + q][v] = cf;
                else if (live && coef_out) *reinterpret_cast<f32x4*>(coef_out + (s0 + sl[q]) * ctot + cb + 4 * v) = cf;
                if (on[q]) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) drow[4 * v + k] = rp[k];
                }
                }
                if (DPTS) {
                    // d feature / d point through the interpolation weights (the ATen grid_sample backward: taps outside the grid contribute
                    // nothing), chained with d coef
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float P0 = tp.wp[0] != 0.f ? plane_val(0, k) : 0.f, P1 = tp.wp[1] != 0.f ? plane_val(1, k) : 0.f;
                        const float P2 = tp.wp[2] != 0.f ? plane_val(2, k) : 0.f, P3 = tp.wp[3] != 0.f ? plane_val(3, k) : 0.f;
                        const float dpx = (P1 - P0) * sn + (P3 - P2) * nn, dpy = (P2 - P0) * ee + (P3 - P1) * ww;
                        const float dl = (tp.wl[1] != 0.f ? line_val(1, k) : 0.f) - (tp.wl[0] != 0.f ? line_val(0, k) : 0.f);
                        gx += dc[k] * lv[k] * dpx;
                        gy += dc[k] * lv[k] * dpy;
                        gl += dc[k] * pv[k] * dl;
                    }
                }
            }
            if (DPTS) {
                // component i feeds the axes (ax, ay | al) = (0, 1 | 2), (0, 2 | 1), (1, 2 | 0): into axis space, then summed over the quad
                // (four consecutive 8-channel groups of one sample: ng = 12 groups are three whole quads, items and lanes are quad-aligned)
                // in registers -- a quarter of the partial sums go through LDS
                const int i = comp[q];
                const float a0 = on[q] ? gx * tp.kx : 0.f, a1 = on[q] ? gy * tp.ky : 0.f, a2 = on[q] ? gl * tp.kl : 0.f;
                float vx = i == 2 ? a2 : a0, vy = i == 0 ? a1 : (i == 1 ? a2 : a0), vz = i == 0 ? a2 : a1;
                vx += dpp_f32<0xb1>(0.f, vx); vy += dpp_f32<0xb1>(0.f, vy); vz += dpp_f32<0xb1>(0.f, vz);
                vx += dpp_f32<0x4e>(0.f, vx); vy += dpp_f32<0x4e>(0.f, vy); vz += dpp_f32<0x4e>(0.f, vz);
                if (on[q] && (lane & 3) == 0) {
                    float* dp = dpart + (sl[q] * 3 + (grp[q] >> 2)) * 3;
                    dp[0] = vx; dp[1] = vy; dp[2] = vz;
                }
            }
        }
    }
#endif
    wave_sync();
    EVD_VBW_T(2);
    // the A operand of the basis gradient's MFMAs (d out[sample 2 u + kb][f = lane & 31]) is fetched HERE, in front of the plane taps'
    // atomics: the VM counter retires in order, a load issued behind them waits for every one of them (stamps: the 24 MFMAs of phase 5 took
    // 14 k cycles with their eight loads issued one by one behind the atomics, a fifth of the tile)
    float bav[VBW_SAMPLES / 2];
#ifdef EVD_VBW_ATOMICS_ONLY
    constexpr bool BASX = false;
#else
    constexpr bool BASX = BAS;
#endif
    if (BASX) {
        const int mn = lane & 31, kb = lane >> 5;
#pragma unroll
        for (int u = 0; u < VBW_SAMPLES / 2; ++u) {
            const long sa = s0 + 2 * u + kb;
            bav[u] = (sa < n && mn < F) ? d_out[sa * (long)d_stride + d_col + mn] : 0.f;
        }
        // ... and waited for here (an L2 hit: the rows were read for phase 1): hipcc cannot count the atomics of the loops below, at the
        // MFMAs it would wait for vmcnt(0).  (Also tried: the NEXT tile's d out / point loads issued here as well -- 0.556 ms either way:
        // the kernel runs at the rate its atomics retire, a wait moved is not a wait removed.)
#pragma unroll
        for (int u = 0; u < VBW_SAMPLES / 2; ++u) asm volatile("" : "+v"(bav[u]));
    }
    if (ISS) {                                    // hand the tile's plane taps to the issuer: cells and weights next to the rows phase 2 wrote
        if (lane < VBW_SAMPLES * 3) {
            const VbwTaps& tp = taps[lane];
#pragma unroll
            for (int t = 0; t < 4; ++t) { hcell[lane * 4 + t] = tp.ip[t]; hwgt[lane * 4 + t] = tp.wp[t]; }
        }
        lds_done();
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) flags[wv] = 1;
    }
    // phase 3: plane taps.  dco now holds the plane rows d coef lv.
    int coff = 0;
#pragma unroll 1
    for (int i = 0; i < 3 && !ISS; ++i) {
        const int C = sel3(i, c0n, c1n, c2n);
        float* gp = sel3(i, gg.plane[0], gg.plane[1], gg.plane[2]);
        if (gp) {
            // lanes = (tap, channel): 64 / C taps of the component per pass (one for the 64-channel x-y plane, all four for a 16-channel
            // plane).  Every lane walks the tile's 16 samples with the sum of a RUN of samples on one cell in a register and adds it once
            // per run: the x-y cell of an NDC ray changes every ~10 samples, and where the importance samples cluster at a surface the
            // x-z / y-z cells repeat as well
            const int tpp = 64 / C < 4 ? 64 / C : 4, j = lane / C, c = lane % C;
            if (C == 64) {
                vbw_walk_plane64(gp, lane, [&](int sm) { const vbw_f32x2 a = *reinterpret_cast<const vbw_f32x2*>(taps[sm * 3 + i].wp), b = *reinterpret_cast<const vbw_f32x2*>(taps[sm * 3 + i].wp + 2);
                                                         return f32x4{a[0], a[1], b[0], b[1]}; },
                                 [&](int sm, int t) { return taps[sm * 3 + i].ip[t]; }, [&](int sm) { return dco[sm * VBW_CSTR + coff + lane]; });
            } else {
#pragma unroll 1
                for (int t0 = 0; t0 < 4; t0 += tpp) {
                    const int t = t0 + (j < tpp ? j : 0);
                    vbw_walk_pass(gp, c, j < tpp, [&](int sm) { return taps[sm * 3 + i].wp[t]; }, [&](int sm) { return taps[sm * 3 + i].ip[t]; },
                                  [&](int sm) { return dco[sm * VBW_CSTR + coff + c]; });
                }
            }
        }
        coff += C;
    }
    if (LINES12) {
        // line taps of components 1 and 2: lanes = (tap, channel of comp 1 | comp 2), 2 x (n_comp[1] + n_comp[2]) <= 64 lanes, one walk
        const int c12 = c1n + c2n, tq = lane / c12, cc = lane % c12;
        if (lane < 2 * c12) {
            const int i = cc < c1n ? 1 : 2, c = cc < c1n ? cc : cc - c1n;
            float* gl = i == 1 ? gg.line[1] : gg.line[2];
            float acc = 0.f;
            bool any = false;
#pragma unroll 4
            for (int s = 0; s < VBW_SAMPLES; ++s) {
                const VbwTaps& tp = taps[s * 3 + i];
                const float w = tp.wl[tq];
                const int cell = tp.il[tq];
                if (w != 0.f) { acc += w * lrow[s * VBW_LROW + cc]; any = true; }
                const bool flush = s == VBW_SAMPLES - 1 || taps[(s + 1) * 3 + i].il[tq] != cell;
                if (flush) {
                    if (any && gl) unsafeAtomicAdd(gl + cell + c, acc);
                    acc = 0.f;
                    any = false;
                }
            }
        }
    }
    EVD_VBW_T(3);
    // phase 4: the point gradient of (sample, axis): the three quads' partial sums
    if (DPTS && lane < VBW_SAMPLES * 3) {
        const int sl = lane / 3, a = lane % 3;
        float sum = 0.f;
        for (int qd = 0; qd < (ng + 3) / 4; ++qd) sum += dpart[(sl * 3 + qd) * 3 + a];
        if (s0 + sl < n) d_pts[(s0 + sl) * 3 + a] = sum;
    }
    EVD_VBW_T(4);
    if (BASX) {
        // phase 5: d basis_mat += d out^T . coef over the tile's 16 samples (coefficient rows: written to the slice by phase 2; the
        // LINES12 form kept them in registers and puts them where the consumed plane rows were)
        const float* crows = CF_LDS ? cfl : dco;
        if (!CF_LDS) {
            wave_sync();
#pragma unroll
            for (int q = 0; q < TRIPS * UNR; ++q) {
                const int t = lane + q * 64;
                if (t < items) {
                    float* crow = dco + (t / ng) * VBW_CSTR + (t % ng) * 8;
#pragma unroll
                    for (int v = 0; v < 2; ++v)
#pragma unroll
                        for (int k = 0; k < 4; ++k) crow[4 * v + k] = cfk[q][v][k];
                }
            }
            wave_sync();
        }
        const int mn = lane & 31, kb = lane >> 5;
#pragma unroll
        for (int u = 0; u < VBW_SAMPLES / 2; ++u) {
            const float av = bav[u];
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                const float bv = 32 * c + mn < ctot ? crows[(2 * u + kb) * (CF_LDS ? CFSTR : VBW_CSTR) + 32 * c + mn] : 0.f;
                bacc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, bacc[c], 0, 0, 0);
            }
        }
    }
    wave_sync();                                  // the slice is rewritten by the next tile
    EVD_VBW_T(5);
    }
    if (ISS && wv < NCW) {                        // the last tile has been taken; tell the issuer this producer is finished
        while (__builtin_amdgcn_readfirstlane(flags[wv]) != 0) __builtin_amdgcn_s_sleep(2);
        if (lane == 0) flags[wv] = 2;
    }
#ifdef EVD_VBW_STAMP
    if (lane == 0 && rows_l) {
        float* o = rows_l + ((long)blockIdx.x * VBW_WAVES + wv) * 8;
        for (int i = 0; i < 6; ++i) o[i] = (float)tph[i];
        o[6] = -7.f; o[7] = (float)((wtiles - ((long)blockIdx.x * VBW_WAVES + wv) + (long)gridDim.x * VBW_WAVES - 1) / ((long)gridDim.x * VBW_WAVES));
    }
#endif
    if (lmax) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) rmaxv = fmaxf(rmaxv, __shfl_xor(rmaxv, o));
        const unsigned mb = __float_as_uint(rmaxv);                    // (non-negative floats order like their bit patterns; +inf above all)
        if (lane == 0 && mb > *reinterpret_cast<volatile unsigned*>(lmax)) atomicMax(lmax, mb);
    }
    if (BAS && gg.basis) {
        // the block's four wavefronts fold their sums through LDS (the basis_mat image is no longer needed), then ONE atomic flush per block
        const int mn = lane & 31, kb = lane >> 5;
        __syncthreads();
        for (int w = 0; w < VBW_WAVES; ++w) {
            if (wv == w && w < NCW) {
#pragma unroll
                for (int c = 0; c < NCT; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int f = (r & 3) + 8 * (r >> 2) + 4 * kb, ch = 32 * c + mn;
                        if (ch < ctot) bs[f * VBW_BSTR + ch] = w == 0 ? bacc[c][r] : bs[f * VBW_BSTR + ch] + bacc[c][r];
                    }
            }
            __syncthreads();
        }
        for (int o = threadIdx.x; o < F * ctot; o += 64 * VBW_WAVES) {
            const int f = o / ctot, ch = o % ctot;
            const float v = bs[f * VBW_BSTR + ch];
            if (v != 0.f) unsafeAtomicAdd(gg.basis + o, v);
        }
    }
}

// d basis_mat[f][c] += sum_s d out[s][f] coef[s][c] over all samples: exact-float32 MFMA (32 x 32 x 2 per sample pair and 32-channel
// tile), every wavefront a strided share of the sample pairs, one atomic flush per wavefront.  HBM-bound: it reads d out and the
// coefficient rows once (512 B per sample).
template <int NCT>
__global__ __launch_bounds__(256) void k_basis_grad(const float* __restrict__ d_out, int d_stride, int d_col, const float* __restrict__ coef, long n, int ctot, int F,
                                                    float* __restrict__ d_basis) {
    const int lane = threadIdx.x & 63, mn = lane & 31, kb = lane >> 5;
    const long gw = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4, pairs = (n + 1) / 2;
    f32x16 acc[NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    constexpr int UN = 4;
    for (long p0 = gw * UN; p0 < pairs; p0 += nw * UN) {
        float a[UN], b[UN][NCT];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long s = 2 * (p0 + u) + kb;
            const bool ok = p0 + u < pairs && s < n;
            a[u] = (ok && mn < F) ? d_out[s * (long)d_stride + d_col + mn] : 0.f;
#pragma unroll
            for (int c = 0; c < NCT; ++c) b[u][c] = (ok && 32 * c + mn < ctot) ? coef[s * ctot + 32 * c + mn] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int c = 0; c < NCT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u][c], acc[c], 0, 0, 0);
    }
    // the block's four wavefronts fold their sums in LDS (plain stores + one barrier each, no LDS atomics), then ONE atomic flush per block:
    // the grid fills every wavefront slot of the chip (the loads are 4-byte lane loads: bandwidth comes from the number of wavefronts)
    __shared__ float fold[32 * 97];
    const int wv = threadIdx.x >> 6;
    for (int w = 0; w < 4; ++w) {
        if (wv == w) {
#pragma unroll
            for (int c = 0; c < NCT; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = (r & 3) + 8 * (r >> 2) + 4 * kb, ch = 32 * c + mn;
                    fold[f * 97 + ch] = w == 0 ? acc[c][r] : fold[f * 97 + ch] + acc[c][r];
                }
        }
        __syncthreads();
    }
    for (int o = threadIdx.x; o < F * ctot; o += 256) {
        const int f = o / ctot, ch = o % ctot;
        const float v = fold[f * 97 + ch];
        if (v != 0.f) unsafeAtomicAdd(d_basis + o, v);
    }
}

// d (TV_loss_app) / d grid, added into `grad` scaled by d loss (a device scalar) x weight (1e-2 planes | 1e-3 lines) (voxnerf.py:126-130, 306-324):
// reg = 2 (sum dh^2 / count_h + sum dw^2 / count_w)  =>  d reg / d x = 4 ((dh_prev - dh_next) / count_h + (dw_prev - dw_next) / count_w)
__device__ __forceinline__ void tv_bwd_body(const float* __restrict__ x, int H, int W, int C, const float* __restrict__ d_loss, float weight, float* __restrict__ grad,
                                            int block, int nblocks) {
    const float scale = d_loss[0] * weight;
    const long per_row = (long)W * (C / 4), total = per_row * H;
    const float kh = H > 1 ? 4.f * scale / ((float)C * (H - 1) * W) : 0.f;
    const float cw = fmaxf((float)C * H * (W - 1), 1.f), kw = 4.f * scale / cw;
    for (long v = (long)block * 256 + threadIdx.x; v < total; v += (long)nblocks * 256) {
        const int hh = (int)(v / per_row);
        const long r = v % per_row;
        const int wq = (int)(r / (C / 4));
        const float* px = x + v * 4;
        const f32x4 c = *reinterpret_cast<const f32x4*>(px);
        f32x4 gsum = {0.f, 0.f, 0.f, 0.f};
        if (hh > 0) gsum += (c - *reinterpret_cast<const f32x4*>(px - (long)W * C)) * kh;
        if (hh + 1 < H) gsum -= (*reinterpret_cast<const f32x4*>(px + (long)W * C) - c) * kh;
        if (wq > 0) gsum += (c - *reinterpret_cast<const f32x4*>(px - C)) * kw;
        if (wq + 1 < W) gsum -= (*reinterpret_cast<const f32x4*>(px + C) - c) * kw;
        f32x4* gd = reinterpret_cast<f32x4*>(grad + v * 4);
        *gd = *gd + gsum;
    }
}
__global__ __launch_bounds__(256) void k_tv_bwd(const float* __restrict__ x, int H, int W, int C, const float* __restrict__ d_loss, float weight, float* __restrict__ grad) {
    tv_bwd_body(x, H, W, C, d_loss, weight, grad, blockIdx.x, gridDim.x);
}
// the six tensors of a level in one launch (voxel.h TvJobs)
__global__ __launch_bounds__(256) void k_tv_bwd_level(const TvJobs jobs, const float* __restrict__ d_loss) {
    int i = 0;
    while (i + 1 < jobs.n && (int)blockIdx.x >= jobs.j[i + 1].blk0) ++i;
    const TvJob jb = jobs.j[i];
    if (jb.grad) tv_bwd_body(jb.x, jb.H, jb.W, jb.C, d_loss, jb.weight, jb.grad, (int)blockIdx.x - jb.blk0, jb.nblk);
}

__global__ __launch_bounds__(256) void k_f32_to_f16(const float* __restrict__ x, long n4, _Float16* __restrict__ y) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256)
        *reinterpret_cast<f16x4*>(y + 4 * i) = __builtin_convertvector(*reinterpret_cast<const f32x4*>(x + 4 * i), f16x4);
}

// ------------------------------------------------------------------------------------------------
// HD hidden width, G geo_feat_dim, FT feature channels read per sample (32 coarse, 64 fine)
template <int PREC, int HD, int G, int FT>
__global__ __launch_bounds__(mlp_threads(PREC), is_half_prec(PREC) ? 2 : 1) void k_voxel_mlp(const VoxMlpParams p) {
    typedef Ops<PREC> O;
    typedef typename O::B B;
    constexpr int NT = mlp_threads(PREC);
    constexpr int T = HD / 32, KS = HD / 16, KF = FT / 16;
    constexpr int FPC = Stream<PREC>::FPC;
    constexpr bool kSmallGeo = (1 + G) <= 32;          // coarse level: [sigma, geo] fits one tile
    constexpr int GT = kSmallGeo ? 1 : G / 32;         // geo tiles
    constexpr int GK = kSmallGeo ? 1 : G / 16;         // geo k-steps fed to the colour net
    static_assert(kSmallGeo || G % 32 == 0, "geo_feat_dim must be < 32 or a multiple of 32");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, h = lane >> 5;
    const long s = (long)blockIdx.x * (NT / 2) + wave * 32 + n;
    const bool valid = s < p.nsamp;
    const long sc = valid ? s : p.nsamp - 1;
    float pts[3], vd[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        pts[c] = p.pts[sc * 3 + c];
        vd[c] = p.viewdirs[(sc / p.S) * p.vd_stride + c];
    }
    // layer-0 input = cat([fts, PE(pts)])  (voxnerf.py:214): FT/16 natural k-steps + the PE arrangement
    B in0[KF + PE_KS], in_dir[PEV_KS];
    {
        const float* f = p.fts + sc * (long)p.ft_stride + 8 * h;
#pragma unroll
        for (int j = 0; j < KF; ++j) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(f + 16 * j), b = *reinterpret_cast<const f32x4*>(f + 16 * j + 4);
            const f32x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
            in0[j] = O::make_b(v);
        }
        B pe[PE_KS];
        encode_b_rt<PREC, PE_KS>(pts, h, p.pe_l, pe);
#pragma unroll
        for (int j = 0; j < PE_KS; ++j) in0[KF + j] = pe[j];
        encode_b_rt<PREC, PEV_KS>(vd, h, p.pe_lv, in_dir);
    }
    const float* lbias = stage_bias<PREC>(smem, p.bias, p.nbias, tid);
    Stream<PREC> st;
    st.start(p.wstream, smem, p.nchunks, tid);
    const float* zero_bias = lbias;        // first 512 floats of the bias block are zeros (sigma net: bias=False)
    const float* cbias = lbias + 32 * 16;

    constexpr int F0 = T * (KF + PE_KS);
    B hid[KS];
    layer<PREC, KF + PE_KS, T, true, OUT_B, 0, false>(st, in0, hid, nullptr, zero_bias, lane, nullptr, HD);
    constexpr int OFF1 = F0 % FPC;
    float sig[16];
    B cin[GK + PEV_KS];
    float* frow = (p.feature && valid) ? p.feature + s * G : nullptr;
    if constexpr (kSmallGeo) {
        // one tile holds [sigma, geo_1..G]; it is both the sigma output and (k-step 0) the colour-net input
        B tmp[2];
        layer<PREC, KS, 1, false, OUT_BOTH, OFF1, false>(st, hid, tmp, sig, zero_bias, lane, nullptr, HD);
        cin[0] = tmp[0];
        if (frow) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row >= 1 && row <= G) frow[row - 1] = sig[r];
            }
        }
    } else {
        layer<PREC, KS, 1, false, OUT_F32, OFF1, false>(st, hid, nullptr, sig, zero_bias, lane, nullptr, HD);
        constexpr int OFF1b = (F0 + KS) % FPC;
        layer<PREC, KS, GT, false, OUT_B, OFF1b, false>(st, hid, cin, nullptr, zero_bias, lane, frow, G);
    }
    constexpr int F1 = kSmallGeo ? KS : KS + GT * KS;
#pragma unroll
    for (int j = 0; j < PEV_KS; ++j) cin[GK + j] = in_dir[j];
    constexpr int OFF2 = (F0 + F1) % FPC;
    B c0[KS], c1[KS];
    layer<PREC, GK + PEV_KS, T, true, OUT_B, OFF2, false>(st, cin, c0, nullptr, cbias, lane, nullptr, HD);
    constexpr int F2 = T * (GK + PEV_KS);
    constexpr int OFF3 = (F0 + F1 + F2) % FPC;
    layer<PREC, KS, T, true, OUT_B, OFF3, false>(st, c0, c1, nullptr, cbias + 32 * T, lane, nullptr, HD);
    constexpr int OFF4 = (F0 + F1 + F2 + T * KS) % FPC;
    float col[16];
    layer<PREC, KS, 1, false, OUT_F32, OFF4, true>(st, c1, nullptr, col, cbias + 64 * T, lane, nullptr, HD);
    if (h == 0 && valid) {
        f32x4 o;
        o[0] = sig[0];
#pragma unroll
        for (int c = 0; c < 3; ++c) o[1 + c] = 1.f / (1.f + expf(-col[c]));      // torch.sigmoid(h) voxnerf.py:252
        *reinterpret_cast<f32x4*>(p.raw + s * 4) = o;
    }
}

template <int PREC, int HD, int G, int FT>
static int launch_vox(const VoxMlpParams& p, hipStream_t st) {
    constexpr int NT = mlp_threads(PREC);
    const long blocks = cdiv(p.nsamp, NT / 2);
    const size_t lds = MlpLds<PREC>::TOTAL;
    EVD_SET_MAX_LDS((&k_voxel_mlp<PREC, HD, G, FT>), lds);
    hipLaunchKernelGGL((k_voxel_mlp<PREC, HD, G, FT>), dim3((unsigned)blocks), dim3(NT), lds, st, p);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

int voxel_mlp_dispatch(int prec, int HD, int G, int FT, const VoxMlpParams& p, hipStream_t st) {
#define EVD_CASE(P, H_, G_, F_) if (prec == P && HD == H_ && G == G_ && FT == F_) return launch_vox<P, H_, G_, F_>(p, st)
    EVD_CASE(EVD_PREC_BF16, 64, 15, 32);
    EVD_CASE(EVD_PREC_F16, 64, 15, 32);
    EVD_CASE(EVD_PREC_F16X3, 64, 15, 32);
    EVD_CASE(EVD_PREC_F32, 64, 15, 32);
    EVD_CASE(EVD_PREC_BF16, 256, 128, 64);
    EVD_CASE(EVD_PREC_F16, 256, 128, 64);
    EVD_CASE(EVD_PREC_F16X3, 256, 128, 64);
    EVD_CASE(EVD_PREC_F32, 256, 128, 64);
#undef EVD_CASE
    return fail(EVD_E_INVALID, "evd_voxel: no kernel for precision %d hidden %d geo %d features %d "
                "(built: coarse 64/15/32, fine 256/128/64)", prec, HD, G, FT);
}

// TVLoss.forward (voxnerf.py:306-324) on a channel-last tensor [H][W][C]; accumulates sum dh^2, sum dw^2.
// HBM-bound (every grid value is read once per training iteration): one thread = 4 channels of one texel, float4
// loads of the texel, its lower and its right neighbour (both re-read from L1/L2), rows strided over blockIdx.y,
// double accumulators, one partial pair per block (summed by k_tv_finish).
__device__ __forceinline__ void tv_body(const float* __restrict__ x, int H, int W, int C, double* __restrict__ acc2, int bxi, int byi, int bx, int by) {
    __shared__ double red[2][4];
    const int vec_per_row = W * (C / 4);
    double sh = 0.0, sw = 0.0;
    for (int hh = byi; hh < H; hh += by) {
        const float* row = x + (long)hh * W * C;
        for (int t = bxi * 256 + threadIdx.x; t < vec_per_row; t += bx * 256) {
            const int wq = t / (C / 4);
            const f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * (long)t);
            float ph = 0.f, pw = 0.f;
            if (hh + 1 < H) {
                const f32x4 d = *reinterpret_cast<const f32x4*>(row + (long)W * C + 4 * (long)t) - v;
                ph = d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
            }
            if (wq + 1 < W) {
                const f32x4 d = *reinterpret_cast<const f32x4*>(row + 4 * (long)t + C) - v;
                pw = d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
            }
            sh += (double)ph;
            sw += (double)pw;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { sh += __shfl_xor(sh, off, 64); sw += __shfl_xor(sw, off, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sh; red[1][threadIdx.x >> 6] = sw; }
    __syncthreads();
    if (threadIdx.x == 0) {         // one partial pair per block (4096 same-address double atomics serialise for ~0.2 ms)
        const int b = byi * bx + bxi;
        acc2[2 * b] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        acc2[2 * b + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}
__global__ __launch_bounds__(256) void k_tv(const float* __restrict__ x, int H, int W, int C, double* __restrict__ acc2) {
    tv_body(x, H, W, C, acc2, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y);
}
__host__ __device__ inline int tv_bx(int W, int C) { const long v = ((long)W * (C / 4) + 255) / 256; return (int)(v < 64 ? v : 64); }
__host__ __device__ inline int tv_by(int H) { return H < 64 ? H : 64; }
// the six tensors of a level in one launch: job i's partial pairs at acc + i * 2 * TV_MAX_BLOCKS, as k_tv_finish reads them
__global__ __launch_bounds__(256) void k_tv_level(const TvJobs jobs, double* __restrict__ acc) {
    int i = 0;
    while (i + 1 < jobs.n && (int)blockIdx.x >= jobs.j[i + 1].blk0) ++i;
    const TvJob jb = jobs.j[i];
    const int lb = (int)blockIdx.x - jb.blk0, bx = tv_bx(jb.W, jb.C), by = tv_by(jb.H);
    tv_body(jb.x, jb.H, jb.W, jb.C, acc + (size_t)i * 2 * TV_MAX_BLOCKS, lb % bx, lb / bx, bx, by);
}

__global__ __launch_bounds__(256) void k_tv_finish(const double* __restrict__ part, TvShape s, float* __restrict__ out) {
    // total = sum_i reg(plane_i) * 1e-2 + reg(line_i) * 1e-3,  reg = 2 (h_tv / count_h + w_tv / count_w)  (voxnerf.py:126-130)
    __shared__ double red[2][4];
    double total = 0.0;
    for (int i = 0; i < 6; ++i) {
        const double* p = part + (long)i * 2 * TV_MAX_BLOCKS;
        double sh = 0.0, sw = 0.0;
        for (int b = threadIdx.x; b < s.blocks[i]; b += blockDim.x) { sh += p[2 * b]; sw += p[2 * b + 1]; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { sh += __shfl_xor(sh, off, 64); sw += __shfl_xor(sw, off, 64); }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sh; red[1][threadIdx.x >> 6] = sw; }
        __syncthreads();
        const double h_tv = red[0][0] + red[0][1] + red[0][2] + red[0][3], w_tv = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        const double ch = (double)s.C[i] * (s.H[i] - 1) * s.W[i];
        double cw = (double)s.C[i] * s.H[i] * (s.W[i] - 1);
        if (cw < 1.0) cw = 1.0;
        total += 2.0 * (h_tv / ch + w_tv / cw) * (i < 3 ? 1e-2 : 1e-3);
    }
    if (threadIdx.x == 0) out[0] = (float)total;
}

int launch_points(const float* rb, int nc, const float* z, long n, int S, float* pts, hipStream_t st) {
    k_points<<<cdiv(n, 256), 256, 0, st>>>(rb, nc, z, n, S, pts);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

int launch_merge_features(const float* old, const float* fresh, const int* order, long R, int S, int N, int F, float* out, int out_stride,
                          hipStream_t st) {
    const long n = R * (long)(S + N) * (F / 4);
    k_merge_features<<<cdiv(n, 256), 256, 0, st>>>(old, fresh, order, R, S, N, F, out, out_stride);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

int launch_merge_features_bwd(const float* d_out, int d_stride, const int* order, long R, int S, int N, int F, float* d_old, float* d_fresh, hipStream_t st) {
    const long n = R * (long)(S + N) * (F / 4);
    k_merge_features_bwd<<<cdiv(n, 256), 256, 0, st>>>(d_out, d_stride, order, R, S, N, F, d_old, d_fresh);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

int launch_voxel_sample(const GridParams& g, bool half_grids, const float* pts, long n, float* out, int out_stride, int out_col, hipStream_t st) {
    const bool wide = (g.n_comp[0] % 8 == 0) && (g.n_comp[1] % 8 == 0) && (g.n_comp[2] % 8 == 0);
    static const bool no_w = env_flag("EVD_VS_BLOCK");      // developer switch: the block-cooperative kernel
    if (wide && g.app_dim <= 32 && !no_w) {
        const unsigned blocks = (unsigned)cdiv(n, (long)VW_SAMPLES * VW_WAVES);
        const int ct = g.n_comp[0] + g.n_comp[1] + g.n_comp[2];
        const size_t lds = vw_basis_bytes(ct) + VW_WAVES * vw_slice_bytes(ct);
        // float32 grids: 174 registers by default = two blocks per CU; compiled for three (168 registers, 5 spilled) -- EVD_VW_F32_OCC=2 selects the former
        static const bool occ2 = []{ const char* e = getenv("EVD_VW_F32_OCC"); return e && e[0] == '2'; }();
        // round 6: k_voxel_sample_m (coefficients straight into the float16 matrix core's operand layout); EVD_GATHER_FORM=w: rounds 3-5's kernel
        static const bool form_w = []{ const char* e = getenv("EVD_GATHER_FORM"); return e && e[0] == 'w'; }();
        long pmax_h = 0;
        for (int i = 0; i < 3; ++i) {
            const long pe = (long)g.grid[i == 2 ? 1 : 0] * g.grid[i == 0 ? 1 : 2] * g.n_comp[i];
            pmax_h = pe > pmax_h ? pe : pmax_h;
        }
        // float16 grids only: on float32 grids (the float32-grade levels) the new form measures equal (67.1 vs 67.0 us) and the old kernel's
        // float32 matrix product is exact -- it stays
        if (half_grids && !form_w && ct % 32 == 0 && ct <= 96 && pmax_h < (1L << 31)) {
            int cus = 256;
            { int dev = 0, v = 0; if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v; }
            const long tiles = cdiv(n, 16L * VM_WAVES), cap = 8L * cus;
            const unsigned mb = (unsigned)(tiles < cap ? tiles : cap);
#ifndef EVD_VM_OCC
#define EVD_VM_OCC 3
#endif
            // (three wavefronts per SIMD, 164 registers, no spills; compiled for four -- 128 registers, 34 spilled -- it runs 70 instead of 57 us)
            k_voxel_sample_m<true, EVD_VM_OCC><<<mb, 64 * VM_WAVES, 0, st>>>(g, pts, n, out, out_stride, out_col);
            EVD_LAUNCH_CHECK();
            return EVD_OK;
        }
        if (half_grids) k_voxel_sample_w<true, 4><<<blocks, 64 * VW_WAVES, lds, st>>>(g, pts, n, out, out_stride, out_col);
        else if (occ2) k_voxel_sample_w<false, 2><<<blocks, 64 * VW_WAVES, lds, st>>>(g, pts, n, out, out_stride, out_col);
        else k_voxel_sample_w<false, 3><<<blocks, 64 * VW_WAVES, lds, st>>>(g, pts, n, out, out_stride, out_col);
        EVD_LAUNCH_CHECK();
        return EVD_OK;
    }
    if (half_grids && wide) k_voxel_sample<true, 8><<<cdiv(n, VS_SAMPLES), 256, 0, st>>>(g, pts, n, out, out_stride, out_col);
    else if (half_grids) k_voxel_sample<true, 4><<<cdiv(n, VS_SAMPLES), 256, 0, st>>>(g, pts, n, out, out_stride, out_col);
    else if (wide) k_voxel_sample<false, 8><<<cdiv(n, VS_SAMPLES), 256, 0, st>>>(g, pts, n, out, out_stride, out_col);
    else k_voxel_sample<false, 4><<<cdiv(n, VS_SAMPLES), 256, 0, st>>>(g, pts, n, out, out_stride, out_col);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

// Persistent blocks of the scatter's main kernel.  The 96-channel MFMA instantiation keeps three blocks per CU: exactly that many blocks
// (768 on the 256 CUs of an MI355X), each walking its share of the tiles, measured best -- 0.98 / 0.79 ms per 2^19 samples (rays along z /
// oblique) against 1.02 / 0.82 with 3072 blocks, 1.14 / 0.94 with 1024 (a ragged last round) and 1.07 / 0.97 with 512: every block pays
// for its basis_mat column and flushes its basis_mat gradient (192 atomic requests) once.  EVD_SCATTER_BLOCKS overrides.
static long scatter_blocks_cap(bool three_per_cu) {
    const char* e = getenv("EVD_SCATTER_BLOCKS");
    const long v = e ? atol(e) : 0;
    if (v > 0) return v;
    if (!three_per_cu) return 3072;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 3072;
    return 3L * cus;
}

int launch_voxel_sample_bwd(const GridParams& g, const float* pts, long n, const float* d_out, int d_stride, int d_col, const GridGrads& gg,
                            float* d_pts, hipStream_t st) {
    if (g.app_dim > VSB_MAXF) return fail(EVD_E_INVALID, "evd_voxel_sample_bwd: app_dim %d > %d", g.app_dim, VSB_MAXF);
    const long tiles = cdiv(n, VS_SAMPLES);
    const bool mm = g.app_dim == 32 && (g.n_comp[0] + g.n_comp[1] + g.n_comp[2]) % 32 == 0;
    const int ct = g.n_comp[0] + g.n_comp[1] + g.n_comp[2];
    const long cap = scatter_blocks_cap(mm && ct <= 96);
    const unsigned blocks = (unsigned)(tiles < cap ? tiles : cap);
    if (mm && ct <= 96) k_voxel_sample_bwd<0, true, 96><<<blocks, 256, 0, st>>>(g, pts, n, d_out, d_stride, d_col, gg, d_pts, BinOut{});
    else if (mm) k_voxel_sample_bwd<0, true, VS_MAXC><<<blocks, 256, 0, st>>>(g, pts, n, d_out, d_stride, d_col, gg, d_pts, BinOut{});
    else k_voxel_sample_bwd<0, false, VS_MAXC><<<blocks, 256, 0, st>>>(g, pts, n, d_out, d_stride, d_col, gg, d_pts, BinOut{});
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

int launch_voxel_sample_bwd_pass1(const GridParams& g, const float* pts, long n, const float* d_out, int d_stride, int d_col, const GridGrads& gg,
                                  float* d_pts, const BinOut& bo, hipStream_t st) {
    if (g.app_dim > VSB_MAXF) return fail(EVD_E_INVALID, "evd_voxel_sample_bwd: app_dim %d > %d", g.app_dim, VSB_MAXF);
    const long tiles = cdiv(n, VS_SAMPLES);
    const bool mm = g.app_dim == 32 && (g.n_comp[0] + g.n_comp[1] + g.n_comp[2]) % 32 == 0;
    const int ct = g.n_comp[0] + g.n_comp[1] + g.n_comp[2];
    const unsigned blocks = (unsigned)(tiles < 3072 ? tiles : 3072);
    if (mm && ct <= 96) k_voxel_sample_bwd<1, true, 96><<<blocks, 256, 0, st>>>(g, pts, n, d_out, d_stride, d_col, gg, d_pts, bo);
    else if (mm) k_voxel_sample_bwd<1, true, VS_MAXC><<<blocks, 256, 0, st>>>(g, pts, n, d_out, d_stride, d_col, gg, d_pts, bo);
    else k_voxel_sample_bwd<1, false, VS_MAXC><<<blocks, 256, 0, st>>>(g, pts, n, d_out, d_stride, d_col, gg, d_pts, bo);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

// hybrid: plane taps by direct atomics here, line taps left as rows + tap records (bo.rows_l, bo.ltap) for k_scatter_lines
int launch_voxel_sample_bwd_planes(const GridParams& g, const float* pts, long n, const float* d_out, int d_stride, int d_col, const GridGrads& gg,
                                   float* d_pts, const BinOut& bo, hipStream_t st) {
    if (g.app_dim > VSB_MAXF) return fail(EVD_E_INVALID, "evd_voxel_sample_bwd: app_dim %d > %d", g.app_dim, VSB_MAXF);
    const long tiles = cdiv(n, VS_SAMPLES);
    const bool mm = g.app_dim == 32 && (g.n_comp[0] + g.n_comp[1] + g.n_comp[2]) % 32 == 0;
    const int ct = g.n_comp[0] + g.n_comp[1] + g.n_comp[2];
    const long cap = scatter_blocks_cap(mm && ct <= 96);
    const unsigned blocks = (unsigned)(tiles < cap ? tiles : cap);
    if (mm && ct <= 96 && bo.rows_p) k_voxel_sample_bwd<3, true, 96><<<blocks, 256, 0, st>>>(g, pts, n, d_out, d_stride, d_col, gg, d_pts, bo);
    else if (mm && ct <= 96) k_voxel_sample_bwd<2, true, 96><<<blocks, 256, 0, st>>>(g, pts, n, d_out, d_stride, d_col, gg, d_pts, bo);
    else if (mm) k_voxel_sample_bwd<2, true, VS_MAXC><<<blocks, 256, 0, st>>>(g, pts, n, d_out, d_stride, d_col, gg, d_pts, bo);
    else k_voxel_sample_bwd<2, false, VS_MAXC><<<blocks, 256, 0, st>>>(g, pts, n, d_out, d_stride, d_col, gg, d_pts, bo);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

// the wavefront-autonomous form (k_voxel_sample_bwd_w + k_basis_grad); the caller runs k_scatter_lines on rows_l / ltap afterwards
bool voxel_sample_bwd_w_ok(const GridParams& g) {
    const int ct = g.n_comp[0] + g.n_comp[1] + g.n_comp[2];
    auto okc = [](int c) { return c == 8 || c == 16 || c == 32 || c == 64; };
    const long pmax = (long)g.grid[0] * g.grid[1] > (long)g.grid[0] * g.grid[2] ? (long)g.grid[0] * g.grid[1] : (long)g.grid[0] * g.grid[2];
    const long pm2 = (long)g.grid[1] * g.grid[2] > pmax ? (long)g.grid[1] * g.grid[2] : pmax;
    // (ct % 32: a sample's 8-channel groups are whole quads of lanes -- the point gradient's quad sums; other widths take the block-cooperative kernel)
    return g.app_dim >= 4 && g.app_dim <= 32 && g.app_dim % 4 == 0 && ct % 32 == 0 && ct <= 96 && okc(g.n_comp[0]) && okc(g.n_comp[1]) && okc(g.n_comp[2]) &&
           pm2 * 64 < (1L << 31) && g.app_act == EVD_ACT_NONE;
}
// OPT-IN (EVD_SCATTER_LINES_INKERNEL=1): the line taps of components 1 and 2 added inside the kernel (run-length walk) when their channels
// fit one pass of 64 lanes.  Measured: 0.725 -> 0.691 ms per 2^19 samples on the micro-benchmark's rays, but the whole blurfactory iteration
// 19.7 -> 20.3 ms: every ray of a batch adds to the same few hundred x / y line cells, and same-address float atomics serialise at the
// memory side (the LDS slices of k_scatter_lines exist for exactly that).
bool voxel_sample_bwd_w_lines12(const GridParams& g) {
    static const bool on = env_flag("EVD_SCATTER_LINES_INKERNEL");
    return on && 2 * (g.n_comp[1] + g.n_comp[2]) <= 64 && g.n_comp[1] + g.n_comp[2] <= 32;
}
int launch_voxel_sample_bwd_w(const GridParams& g, const float* pts, long n, const float* d_out, int d_stride, int d_col, const GridGrads& gg,
                              float* d_pts, float* rows_l, LTap* ltap, float* coef, hipStream_t st, unsigned* lmax, bool half_grids) {
    // EVD_SCATTER_BASIS=separate (developer switch): round 3's form -- coefficient rows to HBM + the k_basis_grad launch
    static const bool bas_sep = [] { const char* e = getenv("EVD_SCATTER_BASIS"); return e && !strcmp(e, "separate"); }();
    const bool bas = gg.basis && !bas_sep;
    int cus = 256;
    { int dev = 0, v = 0; if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v; }
    const bool l12 = voxel_sample_bwd_w_lines12(g);
    // EVD_SCATTER_ISSUER=1 (developer switch, OFF by default): one wavefront of four issues the others' atomics.  Measured with the walk of
    // round 6 (profiles/r06_scatter_issuer_ab.log): 0.444 against 0.433 ms per 2^19 samples -- the kernel is bound by the instructions it
    // issues and, since the walk was cut, by the rate at which atomics retire (17 of 20.5 G requests/s), not by a wavefront's loads waiting
    // behind its own atomics; a quarter fewer compute wavefronts cost what the decoupling buys.
    static const bool iss_on = [] { const char* e = getenv("EVD_SCATTER_ISSUER"); return e && e[0] == '1'; }();
    const bool iss = bas && !l12 && iss_on;
    const long tiles = cdiv(n, (long)VBW_SAMPLES * (iss ? VBI_CW : VBW_WAVES));
    // persistent workgroups with the basis gradient in registers: two per CU (the LDS slices allow no more); else one tile per wavefront
    const unsigned blocks = (unsigned)(bas ? (tiles < 2L * cus ? tiles : 2L * cus) : tiles);
    float* coef_w = (gg.basis && !bas) ? coef : nullptr;
#define EVD_VBW(DP, L12, BAS) { EVD_SET_MAX_LDS((&k_voxel_sample_bwd_w<DP, L12, BAS>), VBW_LDS); \
        k_voxel_sample_bwd_w<DP, L12, BAS><<<blocks, 64 * VBW_WAVES, VBW_LDS, st>>>(g, pts, n, d_out, d_stride, d_col, gg, d_pts, rows_l, ltap, coef_w, lmax); }
#define EVD_VBI(DP) { EVD_SET_MAX_LDS((&k_voxel_sample_bwd_w<DP, false, true, true>), VBI_LDS); \
        k_voxel_sample_bwd_w<DP, false, true, true><<<blocks, 64 * VBW_WAVES, VBI_LDS, st>>>(g, pts, n, d_out, d_stride, d_col, gg, d_pts, rows_l, ltap, coef_w, lmax); }
#define EVD_VBH(DP) { EVD_SET_MAX_LDS((&k_voxel_sample_bwd_w<DP, false, true, false, true>), VBW_LDS); \
        k_voxel_sample_bwd_w<DP, false, true, false, true><<<blocks, 64 * VBW_WAVES, VBW_LDS, st>>>(g, pts, n, d_out, d_stride, d_col, gg, d_pts, rows_l, ltap, coef_w, lmax); }
    // the float16 copies of the grids for the re-gather (the default form of the kernel only): EVD_SCATTER_HALF=0 keeps the float32 grids (A/B)
    static const bool half_on = [] { const char* e = getenv("EVD_SCATTER_HALF"); return !(e && e[0] == '0'); }();
    bool have_h = true;
    for (int i = 0; i < 3; ++i) have_h = have_h && g.plane_h[i] && g.line_h[i];
    if (half_grids && half_on && have_h && bas && !l12 && !iss) {
        if (d_pts) EVD_VBH(true)
        else EVD_VBH(false)
    } else if (iss) {
        if (d_pts) EVD_VBI(true)
        else EVD_VBI(false)
    } else if (bas) {
        if (d_pts && l12) EVD_VBW(true, true, true)
        else if (d_pts) EVD_VBW(true, false, true)
        else if (l12) EVD_VBW(false, true, true)
        else EVD_VBW(false, false, true)
    } else {
        if (d_pts && l12) EVD_VBW(true, true, false)
        else if (d_pts) EVD_VBW(true, false, false)
        else if (l12) EVD_VBW(false, true, false)
        else EVD_VBW(false, false, false)
    }
#undef EVD_VBW
#undef EVD_VBI
#undef EVD_VBH
    EVD_LAUNCH_CHECK();
    if (gg.basis && !bas) {
        const int ct = g.n_comp[0] + g.n_comp[1] + g.n_comp[2];
        const unsigned gb = (unsigned)(cdiv(n, 64L) < 8L * cus ? cdiv(n, 64L) : 8L * cus);     // 8 blocks of 4 wavefronts per CU: every wavefront slot
        if (ct <= 32) k_basis_grad<1><<<gb, 256, 0, st>>>(d_out, d_stride, d_col, coef, n, ct, g.app_dim, gg.basis);
        else if (ct <= 64) k_basis_grad<2><<<gb, 256, 0, st>>>(d_out, d_stride, d_col, coef, n, ct, g.app_dim, gg.basis);
        else k_basis_grad<3><<<gb, 256, 0, st>>>(d_out, d_stride, d_col, coef, n, ct, g.app_dim, gg.basis);
        EVD_LAUNCH_CHECK();
    }
    return EVD_OK;
}

int launch_tv_bwd(const float* x, int H, int W, int C, const float* d_loss, float weight, float* grad, hipStream_t st) {
    const long total = (long)H * W * (C / 4);
    k_tv_bwd<<<(unsigned)(cdiv(total, 256) < 4096 ? cdiv(total, 256) : 4096), 256, 0, st>>>(x, H, W, C, d_loss, weight, grad);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

int launch_f32_to_f16(const float* x, long n, _Float16* y, hipStream_t st) {
    k_f32_to_f16<<<(unsigned)(cdiv(n / 4, 256) < 4096 ? cdiv(n / 4, 256) : 4096), 256, 0, st>>>(x, n / 4, y);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

int launch_tv(const float* x, int H, int W, int C, double* acc2, int* blocks, hipStream_t st) {
    if (C % 4) return fail(EVD_E_INVALID, "evd_voxel_tv_loss: component count %d is not a multiple of 4", C);
    const long vec_per_row = (long)W * (C / 4);
    const unsigned bx = (unsigned)(cdiv(vec_per_row, 256) < 64 ? cdiv(vec_per_row, 256) : 64);
    const unsigned by = (unsigned)(H < 64 ? H : 64);
    *blocks = (int)(bx * by);
    k_tv<<<dim3(bx, by), 256, 0, st>>>(x, H, W, C, acc2);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

int launch_tv_level(TvJobs& jobs, double* partials, TvShape* shape, hipStream_t st) {
    int total = 0;
    for (int i = 0; i < jobs.n; ++i) {
        TvJob& j = jobs.j[i];
        if (j.C % 4) return fail(EVD_E_INVALID, "evd_voxel_tv_loss: component count %d is not a multiple of 4", j.C);
        j.blk0 = total;
        j.nblk = tv_bx(j.W, j.C) * tv_by(j.H);
        shape->C[i] = j.C; shape->H[i] = j.H; shape->W[i] = j.W; shape->blocks[i] = j.nblk;
        total += j.nblk;
    }
    k_tv_level<<<(unsigned)total, 256, 0, st>>>(jobs, partials);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

int launch_tv_bwd_level(TvJobs& jobs, const float* d_loss, hipStream_t st) {
    int total = 0;
    for (int i = 0; i < jobs.n; ++i) {
        TvJob& j = jobs.j[i];
        const long vecs = (long)j.H * j.W * (j.C / 4);
        j.blk0 = total;
        j.nblk = j.grad ? (int)(cdiv(vecs, 256L) < 4096 ? cdiv(vecs, 256L) : 4096) : 0;
        total += j.nblk;
    }
    if (total == 0) return EVD_OK;
    k_tv_bwd_level<<<(unsigned)total, 256, 0, st>>>(jobs, d_loss);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

int launch_tv_finish(const double* acc, const TvShape& s, float* out, hipStream_t st) {
    k_tv_finish<<<1, 256, 0, st>>>(acc, s, out);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

}  // namespace evd
