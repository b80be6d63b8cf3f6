// The WHOLE backward of the 64-wide PDRF level (coarse level of the shipped c2f configs: hidden 64, geo 15, 32 feature channels in;
// reference: autograd of VoxelNeRFBase.forward, networks/pdrf/voxnerf.py:210-254) in ONE launch, the gradient of a 32-sample tile
// RESIDENT IN REGISTERS from d raw to d fts / d PE(pts) (round 5).
//
// The per-layer chain of voxel_train_kernel.h (what the 256-wide level runs) hands every layer's gradient fragments from kernel to kernel
// through HBM and re-reads every activation fragment once for its wgrad and once for its mask: for this level 5 dgrad + 8 wgrad + 8
// reduce launches per backward, ~80 KiB per tile moved for 9 MFLOP -- 1.2 ms of a 15.9 ms blurfactory iteration
// (profiles/r05_train_kernels_before_fused64.txt).  A layer of this level is small enough that ALL its weight gradients fit the register
// file of one wavefront: 19 accumulator blocks of 32 x 32 (304 registers, one wavefront per SIMD).  So a wavefront owns a tile for the
// whole chain:
//   d colour / d sigma fragments from (d raw, raw) in registers (what k_voxel_grad_frags wrote to the store);
//   per layer, top down: wgrad -- the gradient block and the activation block transposed on the matrix core (transpose_block) and
//   multiplied into the layer's accumulators, the bias gradients as columns of ones into one shared block -- then dgrad -- W^T (all 36 fragments of the five
//   layers resident in LDS) times the gradient fragments, masked by [stored activation != 0] -- whose output fragments are the next
//   layer's gradient, never stored;
//   only d PE(dirs), d fts and d PE(pts) leave (the fragments k_pe_bwd / k_frags_to_rows turn into rows).
// Each of the 22 stored activation fragments is read ONCE (22 KiB per tile + 1 KiB of d raw / raw; 8 KiB written), two phases ahead of
// its use with plain 16-byte loads (hipcc counts them: no hand-kept vmcnt).  At the end the four wavefronts of a workgroup fold their
// accumulators through LDS and write one partial block set per workgroup; k_wgrad_reduce_jobs sums them into the five parameter
// gradients in one launch.
#pragma once

#include "nerf_train_kernel.h"
#include "voxel_mlp_kernel.h"
#include "voxel_train.h"

namespace evd {

namespace f64 {
constexpr int HD = 64, G = 15, FT = 32;
typedef VStore<HD, G, FT> VS;
static_assert(VS::KS == 4 && VS::KF == 2 && VS::GT == 1 && PE_KS == 4 && PEV_KS == 2, "fragment counts of the 64-wide level");
static_assert(VS::D_PE == VS::D_FTS + 2 && VS::DIRPE == VS::GEO + 2, "output fragments behind one another");
// accumulator blocks of a wavefront, per layer [row tile][column tile]: the partial layout k_wgrad_reduce reads; the five bias gradients
// (row sums of the gradient blocks of colour_net.2, .1, .0) share ONE block: block c's sums in column c (a column of ones at position c
// as the B operand adds nothing to the other columns)
constexpr int A_C2 = 0, A_C1 = A_C2 + 2, A_C0 = A_C1 + 4, A_SG = A_C0 + 4, A_L0 = A_SG + 2, A_BIAS = A_L0 + 6, NBLK = A_BIAS + 1;
constexpr int B_C2 = 0, B_C1 = 1, B_C0 = 3;       // bias columns: colour_net.2, colour_net.1 (row tiles 0, 1), colour_net.0 (row tiles 0, 1)
// W^T fragments in LDS: per layer [output tile][k-step] (the streams of evd_voxel_api.hip, group = 1)
constexpr int W_C2 = 0, W_C1 = W_C2 + 2, W_C0 = W_C1 + 8, W_SG = W_C0 + 8, W_L0 = W_SG + 6, W_N = W_L0 + 12;
constexpr int LDS_BYTES = W_N * 1024 + 4 * 4096;
}  // namespace f64

struct VoxBwdFusedParams {
    const float *d_raw, *raw;           // [nsamp, 4]
    long nsamp, tiles;
    char* store;
    const char* wt[VBWD_NSTREAMS];
    const unsigned* maxbits;
    float* partial;                     // [gridDim.x][f64::NBLK][64 lanes][16] float32
    float* d_fts;                       // [nsamp, d_fts_stride] float32 out (loss scale removed; 16-byte aligned rows), or null: fragments at D_FTS
    int d_fts_stride;
};

// one stored fragment (one 32 x 16 block) transposed: transpose_block with an all-zero second fragment, one MFMA
template <int PREC> __device__ __forceinline__ void transpose_single(const W4& f0, const W4& sel0, W4 (&t)[2]) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const f32x16 d = mfma_half<PREC>(f0, sel0, zero);
    typename POps<PREC>::B b;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int e = 0; e < 4; ++e) POps<PREC>::template set_pair<false>(b, e, d[8 * q + 2 * e], d[8 * q + 2 * e + 1]);
#pragma unroll
        for (int e = 0; e < 4; ++e) t[q].w[e] = b.w[e];
    }
}

template <int PREC>
__global__ __launch_bounds__(256, 1) void k_voxel_bwd_fused64(const VoxBwdFusedParams p) {
    using namespace f64;
    static_assert(is_half_prec(PREC), "half-precision fragments");
    typedef POps<PREC> O;
    extern __shared__ __attribute__((aligned(16))) char fsm[];
    char* wl = fsm;                                              // [W_N][1 KiB] W^T fragments of the five layers
    float* fold = reinterpret_cast<float*>(fsm + W_N * 1024);    // [4 wavefronts][1024] accumulator exchange of the epilogue
    pipe_fp16_saturate<PREC>();
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), n = lane & 31, h = lane >> 5;
    {
        constexpr int w0[VBWD_NSTREAMS + 1] = {W_C2, W_C1, W_C0, W_SG, W_L0, W_N};
        static_assert(VBWD_C2 == 0 && VBWD_C1 == 1 && VBWD_C0 == 2 && VBWD_SIGGEO == 3 && VBWD_L0 == 4, "stream order");
#pragma unroll
        for (int k = 0; k < VBWD_NSTREAMS; ++k)
            for (int i = tid; i < (w0[k + 1] - w0[k]) * 64; i += 256)
                *reinterpret_cast<f32x4*>(wl + w0[k] * 1024 + i * 16) = *reinterpret_cast<const f32x4*>(p.wt[k] + (long)i * 16);
    }
    __syncthreads();
    auto WT = [&](int f) -> W4 { return *reinterpret_cast<const W4*>(wl + f * 1024 + lane * 16); };
    // transposition selectors (k_wgrad): sel0[kk][n] = (n == kk), sel1[kk][n] = (n == 16 + kk); re-made where they are used (a dozen VALU
    // instructions) instead of living in registers next to 304 accumulators; `ones(c)`: the bias column at position c
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
    const float scale = grad_scale(*p.maxbits, false);
    constexpr long TB = VS::TILE_BYTES;

    auto ld = [&](long t, int slot) -> W4 { return *reinterpret_cast<const W4*>(p.store + t * TB + (long)slot * 1024 + lane * 16); };
    // wgrad of one (gradient block, activation block) pair: both already transposed
    auto prod = [&](f32x16& a, const W4 (&yt)[2], const W4 (&xt)[2]) {
        a = mfma_half<PREC>(yt[0], xt[0], a);
        a = mfma_half<PREC>(yt[1], xt[1], a);
    };
    auto bias = [&](int c, const W4 (&yt)[2]) {
        const W4 o = ones(c);
        acc[A_BIAS] = mfma_half<PREC>(yt[0], o, acc[A_BIAS]);
        acc[A_BIAS] = mfma_half<PREC>(yt[1], o, acc[A_BIAS]);
    };
    // 32 output rows of a dgrad layer (accumulator d) as the two fragments of the next gradient, masked by the stored activations m0, m1
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

    const long nw = (long)gridDim.x * 4;
    long t = (long)blockIdx.x * 4 + wave;
    const long last = p.tiles - 1;
    // the tile's inputs, by the phase that consumes them; loaded two phases ahead
    W4 c1[4], c0[4], gd[4], hid[4], in0[6];
    f32x4 dr, rw;
    auto load1 = [&](long tt) {
        tt = tt < last ? tt : last;
        const long smp = tt * 32 + n;
        const long sc = smp < p.nsamp ? smp : p.nsamp - 1;
        dr = *reinterpret_cast<const f32x4*>(p.d_raw + sc * 4);
        rw = *reinterpret_cast<const f32x4*>(p.raw + sc * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) c1[j] = ld(tt, VS::C1 + j);
    };
    auto load2 = [&](long tt) {
        tt = tt < last ? tt : last;
#pragma unroll
        for (int j = 0; j < 4; ++j) c0[j] = ld(tt, VS::C0 + j);
    };
    if (t < p.tiles) {
        load1(t);
        load2(t);
    }
    for (; t < p.tiles; t += nw) {
        // ---- phase 1: colour_net.2 (+ sigmoid, voxnerf.py:252) -------------------------------------------------------------
#pragma unroll
        for (int j = 0; j < 4; ++j) gd[j] = ld(t, VS::GEO + j);              // geo_0, geo_1 (padding), PE(dirs)_0, PE(dirs)_1
        asm volatile("" ::: "memory");
        W4 gcol, gsig, gc1[4];
        {
            f32x4 g = {0.f, 0.f, 0.f, 0.f};
            if (h == 0 && t * 32 + n < p.nsamp) {
                g = dr;
#pragma unroll
                for (int c = 1; c < 4; ++c) g[c] = g[c] * rw[c] * (1.f - rw[c]);
            }
            typename O::B col, sg;
            O::zero(col);
            O::zero(sg);
            O::template set_pair<false>(col, 0, g[1] * scale, g[2] * scale);
            O::template set_pair<false>(col, 1, g[3] * scale, 0.f);
            O::template set_pair<false>(sg, 0, g[0] * scale, 0.f);
            gcol = __builtin_bit_cast(W4, col);
            gsig = __builtin_bit_cast(W4, sg);
        }
        {
            W4 yt[2], xt[2], sel0, sel1;
            selectors(sel0, sel1);
            transpose_single<PREC>(gcol, sel0, yt);
#pragma unroll
            for (int xb = 0; xb < 2; ++xb) {
                transpose_block<PREC>(c1[2 * xb], c1[2 * xb + 1], sel0, sel1, xt);
                prod(acc[A_C2 + xb], yt, xt);
            }
            bias(B_C2, yt);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const f32x16 d = mfma_half<PREC>(WT(W_C2 + r), gcol, zero16);
                frags(d, gc1[2 * r], gc1[2 * r + 1]);
                mask(gc1[2 * r], c1[2 * r]);
                mask(gc1[2 * r + 1], c1[2 * r + 1]);
            }
        }
        // ---- phase 2: colour_net.1 ------------------------------------------------------------------------------------------
#pragma unroll
        for (int j = 0; j < 4; ++j) hid[j] = ld(t, VS::HID + j);
        asm volatile("" ::: "memory");
        W4 gc0[4];
        {
            W4 yt[2][2], xt[2], sel0, sel1;
            selectors(sel0, sel1);
#pragma unroll
            for (int yb = 0; yb < 2; ++yb) transpose_block<PREC>(gc1[2 * yb], gc1[2 * yb + 1], sel0, sel1, yt[yb]);
#pragma unroll
            for (int xb = 0; xb < 2; ++xb) {
                transpose_block<PREC>(c0[2 * xb], c0[2 * xb + 1], sel0, sel1, xt);
#pragma unroll
                for (int yb = 0; yb < 2; ++yb) prod(acc[A_C1 + 2 * yb + xb], yt[yb], xt);
            }
#pragma unroll
            for (int yb = 0; yb < 2; ++yb) bias(B_C1 + yb, yt[yb]);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                f32x16 d = zero16;
#pragma unroll
                for (int j = 0; j < 4; ++j) d = mfma_half<PREC>(WT(W_C1 + 4 * r + j), gc1[j], d);
                frags(d, gc0[2 * r], gc0[2 * r + 1]);
                mask(gc0[2 * r], c0[2 * r]);
                mask(gc0[2 * r + 1], c0[2 * r + 1]);
            }
        }
        // ---- phase 3: colour_net.0 on cat([geo, PE(dirs)]) ------------------------------------------------------------------
#pragma unroll
        for (int j = 0; j < 6; ++j) in0[j] = ld(t, VS::IN0 + j);
        asm volatile("" ::: "memory");
        W4 dgeo0;
        {
            W4 yt[2][2], xt[2], sel0, sel1;
            selectors(sel0, sel1);
#pragma unroll
            for (int yb = 0; yb < 2; ++yb) transpose_block<PREC>(gc0[2 * yb], gc0[2 * yb + 1], sel0, sel1, yt[yb]);
#pragma unroll
            for (int xb = 0; xb < 2; ++xb) {
                transpose_block<PREC>(gd[2 * xb], gd[2 * xb + 1], sel0, sel1, xt);
#pragma unroll
                for (int yb = 0; yb < 2; ++yb) prod(acc[A_C0 + 2 * yb + xb], yt[yb], xt);
            }
#pragma unroll
            for (int yb = 0; yb < 2; ++yb) bias(B_C0 + yb, yt[yb]);
            char* out = p.store + t * TB + lane * 16;
#pragma unroll
            for (int r = 0; r < 2; ++r) {        // output tile 0: d geo (15 rows: its second fragment is zero), tile 1: d PE(dirs)
                f32x16 d = zero16;
#pragma unroll
                for (int j = 0; j < 4; ++j) d = mfma_half<PREC>(WT(W_C0 + 4 * r + j), gc0[j], d);
                W4 o0, o1;
                frags(d, o0, o1);
                if (r == 0) dgeo0 = o0;
                else {
                    *reinterpret_cast<W4*>(out + (long)(VS::D_DIRPE) * 1024) = o0;
                    *reinterpret_cast<W4*>(out + (long)(VS::D_DIRPE + 1) * 1024) = o1;
                }
            }
        }
        // ---- phase 4: sigma_net.1 = [sigma row | geo rows] on hid -------------------------------------------------------------
        if (t + nw < p.tiles) load1(t + nw);
        asm volatile("" ::: "memory");
        W4 dhid[4];
        {
            W4 yt[2], xt[2], sel0, sel1;
            selectors(sel0, sel1);
            transpose_block<PREC>(dgeo0, gsig, sel0, sel1, yt);          // rows 0..15: geo positions, rows 16..31: the sigma fragment
#pragma unroll
            for (int xb = 0; xb < 2; ++xb) {
                transpose_block<PREC>(hid[2 * xb], hid[2 * xb + 1], sel0, sel1, xt);
                prod(acc[A_SG + xb], yt, xt);
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {        // k-steps of the stream: geo_0, geo_1 (all padding for 15 channels), sigma
                f32x16 d = mfma_half<PREC>(WT(W_SG + 3 * r), dgeo0, zero16);
                d = mfma_half<PREC>(WT(W_SG + 3 * r + 2), gsig, d);
                frags(d, dhid[2 * r], dhid[2 * r + 1]);
                mask(dhid[2 * r], hid[2 * r]);
                mask(dhid[2 * r + 1], hid[2 * r + 1]);
            }
        }
        // ---- phase 5: sigma_net.0 on cat([fts, PE(pts)]) ---------------------------------------------------------------------
        if (t + nw < p.tiles) load2(t + nw);
        asm volatile("" ::: "memory");
        {
            W4 yt[2][2], xt[2], sel0, sel1;
            selectors(sel0, sel1);
#pragma unroll
            for (int yb = 0; yb < 2; ++yb) transpose_block<PREC>(dhid[2 * yb], dhid[2 * yb + 1], sel0, sel1, yt[yb]);
#pragma unroll
            for (int xb = 0; xb < 3; ++xb) {
                transpose_block<PREC>(in0[2 * xb], in0[2 * xb + 1], sel0, sel1, xt);
#pragma unroll
                for (int yb = 0; yb < 2; ++yb) prod(acc[A_L0 + 3 * yb + xb], yt[yb], xt);
            }
            char* out = p.store + t * TB + lane * 16;
#pragma unroll
            for (int r = 0; r < 3; ++r) {        // output tile 0: d fts, tiles 1, 2: d PE(pts)
                f32x16 d = zero16;
#pragma unroll
                for (int j = 0; j < 4; ++j) d = mfma_half<PREC>(WT(W_L0 + 4 * r + j), dhid[j], d);
                W4 o0, o1;
                frags(d, o0, o1);
                if (r == 0 && p.d_fts) {         // the feature gradient leaves as the float32 rows the tri-plane scatter reads (k_frags_to_rows' output)
                    const long smp = t * 32 + n;
                    if (smp < p.nsamp) frag_pair_to_row<PREC>(o0, o1, p.d_fts + smp * p.d_fts_stride, h, grad_scale(*p.maxbits, true));
                    continue;
                }
                *reinterpret_cast<W4*>(out + (long)(VS::D_FTS + 2 * r) * 1024) = o0;
                *reinterpret_cast<W4*>(out + (long)(VS::D_FTS + 2 * r + 1) * 1024) = o1;
            }
        }
    }
    // ---- the workgroup's four accumulator sets folded through LDS: one partial block set per workgroup ----------------------------
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

// the shared bias block: column c of the workgroups' partial blocks summed -> the bias gradient it belongs to (rows through an index map)
struct BiasColsParams {
    const float* partial;               // the bias block of workgroup 0
    int nparts;
    long part_stride;
    int ncols;                          // <= 8 columns in use
    const int* rowmap[8];               // column c: accumulator row nr -> row of db[c] (or -1)
    float* db[8];                       // null: not wanted
    const unsigned* maxbits;
    int accum;
};
// one wavefront per used element (ncols columns x 32 rows): its lanes share the workgroups' partial blocks
static __global__ __launch_bounds__(256) void k_bias_cols_reduce(const BiasColsParams p) {
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e >= p.ncols * 32) return;
    const int nc = e % p.ncols, nr = e / p.ncols;                 // accumulator element (lane', i): column nc = lane' & 31, row nr = (i & 3) + 8 (i >> 2) + 4 (lane' >> 5)
    const int idx = (nc + 32 * ((nr >> 2) & 1)) * 16 + (nr & 3) + 4 * (nr >> 3);
    float s = 0.f;
    for (int q = lane; q < p.nparts; q += 64) s += p.partial[(long)q * p.part_stride + idx];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if (lane != 0) return;
    float* db = p.db[nc];
    if (!db) return;
    const int row = p.rowmap[nc][nr];
    if (row < 0) return;
    const float v = s * grad_scale(*p.maxbits, true);
    db[row] = p.accum ? db[row] + v : v;
}

}  // namespace evd
