// Backward of the fused NeRF MLP (bf16 / f16 modes, netdepth 8, netwidth 256, skips [4]).
// reference: the autograd graph of NeRF.mlpforward, networks/nerf.py:46-72, differentiated by run_nerf.py:593-601.
//
// Everything works on the activation store the training forward filled (nerf_mlp.h, namespace astore): one 1 KiB MFMA
// B fragment per (32-sample tile, 16 channels), lane-linear, so every access below is a coalesced 16-byte load/store
// per lane and no kernel ever re-arranges data through LDS.
//
//   k_grad_frags   d raw [n,4] (float32) x 2^s  ->  the two gradient fragments G_RGB, G_ALPHA (s: power-of-two loss
//                  scale that puts max |d raw| in [512, 1024), so float16 gradients neither overflow nor flush)
//   k_dgrad_layer  d x = (W^T d y) . [x > 0] for ONE layer, on the software pipeline of the forward kernel: W^T is the
//                  streamed A operand, the incoming gradient fragments are the B operand, the outgoing ones are
//                  multiplied by the ReLU pattern of the saved activation in the epilogue and stored
//   k_wgrad        dW = d y . x^T over all samples.  Both operands are stored sample-minor, the contraction runs over
//                  samples: each 32x32 block is first transposed ON THE MATRIX CORE (the stored fragment as the A
//                  operand against a 0/1 selector as B returns it with samples along the accumulator registers),
//                  converted back to half precision (exact) and fed to the product MFMAs.  A wavefront owns RPW row
//                  tiles x all column tiles of dW in accumulators and walks the sample tiles with a grid stride;
//                  partial sums go to a float32 scratch, bias gradients ride along as a column of ones.
//   k_wgrad_reduce sums the partials, undoes the fragment permutations through two index maps, removes the loss scale
#pragma once

#include "mlp_pipe.h"
#include "nerf_train.h"

namespace evd {

// 2^s with max * 2^s in [512, 1024); `bits` = float bits of max |d raw| (0: no gradient at all)
__device__ __forceinline__ float grad_scale(unsigned bits, bool inverse) {
    const float m = __uint_as_float(bits);
    if (!(m > 0.f) || !(m < 3.0e38f)) return 1.f;
    int e;
    (void)frexpf(m, &e);                                 // m = f 2^e, f in [0.5, 1)
    return ldexpf(1.f, inverse ? e - 10 : 10 - e);
}

static __global__ __launch_bounds__(256) void k_absmax(const float* __restrict__ x, long n, unsigned* __restrict__ out) {
    __shared__ float part[4];
    float m = 0.f;
    const long n4 = n >> 2, stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {          // float4 body (the gradient rows are 16-byte aligned)
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    for (long i = 4 * n4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    // one atomic per block: thousands of same-address atomicMax (one per wavefront, as first written) serialise at the L2 and cost
    // more than the read of the whole array
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
        if (m > 0.f) atomicMax(out, __float_as_uint(m));
    }
}

template <int PREC>
__global__ __launch_bounds__(256) void k_grad_frags(const float* __restrict__ d_raw, long nsamp, const unsigned* __restrict__ maxbits,
                                                    char* __restrict__ store, long tiles) {
    typedef POps<PREC> O;
    typedef typename O::B B;
    pipe_fp16_saturate<PREC>();
    const long idx = (long)blockIdx.x * 256 + threadIdx.x, tile = idx >> 6;
    if (tile >= tiles) return;
    const int lane = idx & 63, n = lane & 31, h = lane >> 5;
    const long smp = tile * 32 + n;
    const float s = grad_scale(*maxbits, false);
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    if (h == 0 && smp < nsamp) g = *reinterpret_cast<const f32x4*>(d_raw + smp * 4);
    B rgb, al;
    O::zero(rgb);
    O::zero(al);
    O::template set_pair<false>(rgb, 0, g[0] * s, g[1] * s);
    O::template set_pair<false>(rgb, 1, g[2] * s, 0.f);
    O::template set_pair<false>(al, 0, g[3] * s, 0.f);
    constexpr int FB = frag_bytes(PREC);
    char* a = store + tile * astore::tile_bytes(PREC) + lane * 16;
    act_store<FB>(a, astore::G_RGB, rgb);
    act_store<FB>(a, astore::G_ALPHA, al);
}

template <class B, int FB = 1024> __device__ __forceinline__ B frag_load(const char* lane_base, int slot) {
    static_assert(sizeof(B) == 16 || (sizeof(B) == 32 && FB == 2048), "16-byte fragments, or hi / lo pairs in 2 KiB slots");
    if constexpr (sizeof(B) == 16) {
        return __builtin_bit_cast(B, *reinterpret_cast<const f32x4*>(lane_base + (long)slot * FB));
    } else {
        struct Two { f32x4 a, b; };
        Two t;
        t.a = *reinterpret_cast<const f32x4*>(lane_base + (long)slot * FB);
        t.b = *reinterpret_cast<const f32x4*>(lane_base + (long)slot * FB + 1024);
        return __builtin_bit_cast(B, t);
    }
}

// the 8 values of one lane of a stored fragment as float32 (split mode: hi + lo / 2048)
template <int PREC> __device__ __forceinline__ void frag_values(const char* lane_base, int slot, float (&v)[8]) {
    constexpr int FB = frag_bytes(PREC);
    const MaskFrag f = frag_load<MaskFrag, FB>(lane_base, slot);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const unsigned short bits = (unsigned short)(f.w[e >> 1] >> (16 * (e & 1)));
        if constexpr (PREC == EVD_PREC_BF16) v[e] = __uint_as_float((unsigned)bits << 16);
        else v[e] = (float)__builtin_bit_cast(_Float16, bits);
    }
    if constexpr (PREC == EVD_PREC_F16X3) {
        const MaskFrag l = __builtin_bit_cast(MaskFrag, *reinterpret_cast<const f32x4*>(lane_base + (long)slot * FB + 1024));
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const unsigned short bits = (unsigned short)(l.w[e >> 1] >> (16 * (e & 1)));
            v[e] = fmaf((float)__builtin_bit_cast(_Float16, bits), 4.8828125e-4f, v[e]);
        }
    }
}

struct W4 { unsigned w[4]; };

// The two fragments of a 32-row dgrad output tile (accumulator rows (i & 3) + 8 (i >> 2) + 4 h as pairs, already rounded to half
// precision) written as float32 ROWS: this lane's sample, features 16 f + 8 g + 4 h + 0..3 of the tile for fragment f, half g -- what
// k_frags_to_rows makes of the stored fragments (voxel_train_kernel.h), without the round trip.  row = the sample's row + the tile's
// first feature; 16-byte aligned (checked by the launcher).
template <int PREC> __device__ __forceinline__ void frag_pair_to_row(const W4& f0, const W4& f1, float* row, int h, float inv) {
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const W4& fr = f ? f1 : f0;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            f32x4 v;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned short bits = (unsigned short)(fr.w[2 * g + (k >> 1)] >> (16 * (k & 1)));
                if constexpr (PREC == EVD_PREC_BF16) v[k] = __uint_as_float((unsigned)bits << 16) * inv;
                else v[k] = (float)__builtin_bit_cast(_Float16, bits) * inv;
            }
            *reinterpret_cast<f32x4*>(row + 16 * f + 8 * g + 4 * h) = v;
        }
    }
}

struct DgradParams {
    const char* wstream;    // W^T as a fragment stream (pack.h), one layer
    char* store;
    long tile_bytes;        // stride of a 32-sample tile in the store
    int in_slot, extra_slot, mask_slot, out_slot;
    // optional: max |output| in true units (the loss scale behind *maxbits removed) -> atomicMax on *absmax_out (float bits), one per workgroup
    unsigned* absmax_out = nullptr;
    const unsigned* maxbits = nullptr;
};

// One dgrad layer: KTOT k-steps in (NIN contiguous fragments from in_slot, already masked by their producer, plus one more
// from extra_slot if EXTRA), TILES 32-row tiles out, multiplied by the ReLU pattern of the activation fragments at mask_slot
// (OMASK) before they are stored: what the next dgrad layer and the wgrad kernel read is d loss / d pre-activation.
// 8 wavefronts x 32 samples per workgroup.
template <int PREC, int KTOT, int TILES, int NIN, bool EXTRA, int OMASK, int NT>
__global__ __launch_bounds__(NT, NT / 256) void k_dgrad_layer(const DgradParams p) {
    typedef PipeCfg<PREC, 1, NT> C;
    typedef typename C::O O;
    typedef typename O::B B;
    typedef LayerDesc<KTOT, TILES, 1, false, false, 0, 0, true, 0, 0, 0, false, 0, -1, false, 0, 0, -1> L;
    constexpr int NCH = cceil(KTOT * TILES, C::FPC), FB = C::FB;
    static_assert(NIN + (EXTRA ? 1 : 0) == KTOT, "k-steps of the layer");    // (a layer shorter than the prefetch depth reads the zero padding of its chunk)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    pipe_fp16_saturate<PREC>();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    PStream<C, true, NCH> st;
    st.start_issue(p.wstream, smem, tid);
    float* zb = reinterpret_cast<float*>(smem + C::RING);       // the dgrad layers have no bias: a block of zeros
    for (int i = tid; i < (TILES + 1) * 32; i += NT) zb[i] = 0.f;
    char* al = p.store + ((long)blockIdx.x * (NT / 64) + wave) * p.tile_bytes + lane * 16;

    B in[1][KTOT], out[1][2 * TILES], oma[OMASK == 1 ? 2 * TILES : 1];
    MaskFrag omb[1];
#pragma unroll
    for (int j = 0; j < NIN; ++j) in[0][j] = frag_load<B, FB>(al, p.in_slot + j);
    if constexpr (EXTRA) in[0][NIN] = frag_load<B, FB>(al, p.extra_slot);
    if constexpr (OMASK == 1) {
#pragma unroll
        for (int j = 0; j < 2 * TILES; ++j) oma[j] = frag_load<B, FB>(al, p.mask_slot + j);
    } else if constexpr (OMASK == 2) {
        omb[0] = frag_load<MaskFrag, FB>(al, p.mask_slot);          // the bit-mask fragment (nerf_mlp.h M_H0 ..)
    }
    const void* om = OMASK == 1 ? static_cast<const void*>(oma) : static_cast<const void*>(omb);
    char* actl[1] = {al + (long)p.out_slot * FB};
    float* nofrow[1] = {nullptr};
    Pipe<C> pp;
    st.start_wait();
    pipe_prime<C, L>(st, pp, zb, lane);
    pipe_layer<C, L, decltype(st), 2 * TILES, true, OMASK>(st, pp, in, out, nullptr, zb, lane, nofrow, actl, om);
    pipe_flush<C, L>(pp, out);
#pragma unroll
    for (int j = 2 * TILES - 2; j < 2 * TILES; ++j) {
        if constexpr (OMASK == 1) O::mask_act(out[0][j], oma[j]);
        else if constexpr (OMASK == 2) O::mask_bits(out[0][j], omb[0], j);
        act_store<FB>(actl[0], j, out[0][j]);
    }
    if constexpr (is_half_prec(PREC)) {
        if (p.absmax_out) {              // (what a separate pass over the stored fragments computed: k_frag_absmax, 110 us at 1.3 M samples x 128)
            float m = 0.f;
#pragma unroll
            for (int j = 0; j < 2 * TILES; ++j) {
                const W4 f = __builtin_bit_cast(W4, out[0][j]);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned short bits = (unsigned short)(f.w[e >> 1] >> (16 * (e & 1)));
                    float v;
                    if constexpr (PREC == EVD_PREC_BF16) v = __uint_as_float((unsigned)bits << 16);
                    else v = (float)__builtin_bit_cast(_Float16, bits);
                    m = v != v ? __builtin_huge_valf() : fmaxf(m, fabsf(v));          // (a NaN is recorded as +inf, like k_awp_bwd_fused and k_mam_local_bwd)
                }
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
            __syncthreads();             // (the ring is not read any more: its first words carry the wavefronts' maxima)
            float* part = reinterpret_cast<float*>(smem);
            if (lane == 0) part[wave] = m;
            __syncthreads();
            if (tid == 0) {
                for (int w = 1; w < NT / 64; ++w) m = fmaxf(m, part[w]);
                // (thousands of atomicMax on one word serialise at the L2 -- 75 -> 309 us for this launch --: only a workgroup that would raise
                // the word as it reads it now sends one; positive floats order like their bit patterns)
                const unsigned mb = __float_as_uint(m * grad_scale(*p.maxbits, true));
                if (m > 0.f && mb > *reinterpret_cast<volatile unsigned*>(p.absmax_out)) atomicMax(p.absmax_out, mb);
            }
        }
    }
}

template <int PREC, int KTOT, int TILES, int NIN, bool EXTRA, int OMASK>
static int launch_dgrad(const DgradParams& p, long tiles, hipStream_t st) {
    constexpr int NT = is_half_prec(PREC) ? 512 : 256;          // the split mode's fragments need the whole register file: one wavefront per SIMD
    typedef PipeCfg<PREC, 1, NT> C;
    const size_t lds = C::RING + (TILES + 1) * 128;
    EVD_SET_MAX_LDS((&k_dgrad_layer<PREC, KTOT, TILES, NIN, EXTRA, OMASK, NT>), lds);
    hipLaunchKernelGGL((k_dgrad_layer<PREC, KTOT, TILES, NIN, EXTRA, OMASK, NT>), dim3((unsigned)(tiles / (NT / 64))), dim3(NT), lds, st, p);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

// ------------------------------------------------------------------------------------------------
struct WgradParams {
    const char* store;
    long tiles, tile_bytes;
    int y_slot, x_slot, bias;       // bias: the bias gradient rides along as column CT
    float* partial;         // [gridDim.x][RT][CT + BIAS][64][16] float32
};

constexpr int WGRAD_NT = 512, WGRAD_TPI = 1;      // 8 wavefronts; sample tiles per iteration (one barrier each)
constexpr int wgrad_cpg(int RT, int CT) { return cceil(CT, 8 / RT); }      // column tiles per wavefront group


template <int PREC> __device__ __forceinline__ f32x16 mfma_half(const W4& a, const W4& b, const f32x16& c) {
    if constexpr (PREC == EVD_PREC_BF16) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// the 32-channel x 32-sample block held by fragments (f0, f1), returned sample-major: operand q (16 samples) of a
// lane = channel column (lane & 31) of the block; both MFMA operand roles read it the same way
template <int PREC> __device__ __forceinline__ void transpose_block(const W4& f0, const W4& f1, const W4& sel0, const W4& sel1, W4 (&t)[2]) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 d = mfma_half<PREC>(f0, sel0, zero);
    d = mfma_half<PREC>(f1, sel1, d);
    typename POps<PREC>::B b;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int e = 0; e < 4; ++e) POps<PREC>::template set_pair<false>(b, e, d[8 * q + 2 * e], d[8 * q + 2 * e + 1]);
#pragma unroll
        for (int e = 0; e < 4; ++e) t[q].w[e] = b.w[e];
    }
}

template <int PREC> __device__ __forceinline__ unsigned half_one_pair() { return PREC == EVD_PREC_BF16 ? 0x3f803f80u : 0x3c003c00u; }

// RT row tiles (fragments y_slot .. ; YSINGLE: one fragment, 16 channels) x CT column tiles (fragments x_slot ..).
// The 8 wavefronts of a workgroup share every sample tile: wavefront w owns row tile w % RT and the column tiles of its
// group w / RT in accumulators; it transposes its own gradient block, and (w < CT) the activation block of column tile w,
// which it publishes through LDS for the others.  The stored fragments arrive by LDS-DMA (one global_load_lds_dwordx4 = one
// 1 KiB fragment, no registers in flight) into a ring of WG_RING tiles per wavefront, WG_RING - 1 tiles ahead of their use,
// with counted vmcnt waits; only the owning wavefront reads its slots, so the ring needs no barrier.  (Measured: a ring of 4 with a
// single-buffered exchange runs at the same speed; with the products or the barrier ablated, EVD_ABL_WG_*, the kernel is no faster:
// it is bound by its stream of fragment loads at ~3.7 TB/s.)
constexpr int WG_RING = 3;
constexpr int wgrad_lds_bytes(int CT) { return WG_RING * 8 * 4 * 1024 + 2 * CT * 2 * 1024; }

template <int PREC, int RT, int CT, bool YSINGLE>
__global__ __launch_bounds__(WGRAD_NT) void k_wgrad(const WgradParams p) {
    constexpr int CPG = wgrad_cpg(RT, CT);
    static_assert(8 % RT == 0 && CT <= 8 && (!YSINGLE || RT == 1), "shape of the block");
    extern __shared__ __attribute__((aligned(16))) char wsm[];
    char* raw = wsm;                                          // [WG_RING][8 wavefronts][4 fragments][1 KiB]
    char* xs = wsm + WG_RING * 8 * 4 * 1024;                  // [2][CT][2][1 KiB] transposed activation blocks
    const int NC = CT + (p.bias ? 1 : 0);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n = lane & 31, h = lane >> 5;
    const int rt = wave % RT, c0 = (wave / RT) * CPG;
    const bool xown = wave < CT, bias_own = p.bias && wave < RT;
    // selectors: sel0[kk][n] = (n == kk), sel1[kk][n] = (n == 16 + kk); this lane holds kk = 8h .. 8h + 7 of column n
    W4 sel0, sel1, ones;
    const unsigned one = half_one_pair<PREC>();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int kk = 8 * h + 2 * e;
        sel0.w[e] = (n == kk ? (one & 0xffffu) : 0u) | (n == kk + 1 ? (one & 0xffff0000u) : 0u);
        sel1.w[e] = (n == 16 + kk ? (one & 0xffffu) : 0u) | (n == 17 + kk ? (one & 0xffff0000u) : 0u);
        ones.w[e] = n == 0 ? one : 0u;
    }
    const W4 zf = {{0u, 0u, 0u, 0u}};
    f32x16 acc[CPG], accb;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        accb[i] = 0.f;
#pragma unroll
        for (int c = 0; c < CPG; ++c) acc[c][i] = 0.f;
    }
    // this wavefront's four fragments of a tile: y0, y1 (row tile rt), x0, x1 (column tile `wave`); absent ones (single-fragment
    // gradients, wavefronts beyond CT) re-load y0 so that every tile costs the same four DMA instructions (vmcnt accounting)
    const long oy0 = (long)(p.y_slot + 2 * rt) * 1024, oy1 = YSINGLE ? oy0 : oy0 + 1024;
    const long ox0 = xown ? (long)(p.x_slot + 2 * wave) * 1024 : oy0, ox1 = xown ? ox0 + 1024 : oy0;
    const unsigned ring0 = lds_offset_of(raw) + wave * 4096;
    auto issue = [&](long t, int stage) {
#ifdef EVD_ABL_WG_NODMA
        return;
#endif
        if (t >= p.tiles) return;
        const char* g = p.store + t * p.tile_bytes + lane * 16;
        const unsigned dst = __builtin_amdgcn_readfirstlane(ring0 + stage * (8 * 4096));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %2, off offset:1024\n\t"
                     "global_load_lds_dwordx4 %3, off offset:2048\n\tglobal_load_lds_dwordx4 %4, off offset:3072\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g + oy0), "v"(g + oy1 - 1024), "v"(g + ox0 - 2048), "v"(g + ox1 - 3072), "s"(dst) : "memory");
    };
    const long stride = gridDim.x;
    long t = blockIdx.x;
#pragma unroll
    for (int k = 0; k < WG_RING - 1; ++k) issue(t + k * stride, k);
    for (int it = 0; t < p.tiles; t += stride, ++it) {
        issue(t + (WG_RING - 1) * stride, (it + WG_RING - 1) % WG_RING);
        // the tile's four loads have landed when at most (tiles issued after it) x 4 are still outstanding
        const int later = (t + stride < p.tiles ? 1 : 0) + (t + 2 * stride < p.tiles ? 1 : 0);
        if (later == 2) wait_vmcnt<8>();
        else if (later == 1) wait_vmcnt<4>();
        else wait_vmcnt<0>();
        const char* rw = raw + ((it % WG_RING) * 8 + wave) * 4096 + lane * 16;
        const W4 y0 = *reinterpret_cast<const W4*>(rw), y1 = YSINGLE ? zf : *reinterpret_cast<const W4*>(rw + 1024);
        W4 yt[2];
        transpose_block<PREC>(y0, y1, sel0, sel1, yt);
        char* xb = xs + (it & 1) * (CT * 2048);
        if (xown) {
            const W4 x0 = *reinterpret_cast<const W4*>(rw + 2048), x1 = *reinterpret_cast<const W4*>(rw + 3072);
            W4 xt[2];
            transpose_block<PREC>(x0, x1, sel0, sel1, xt);
#pragma unroll
            for (int q = 0; q < 2; ++q) *reinterpret_cast<W4*>(xb + (wave * 2 + q) * 1024 + lane * 16) = xt[q];
        }
#ifndef EVD_ABL_WG_NOBAR
        __syncthreads();
#endif
#ifndef EVD_ABL_WG_NOPROD
#pragma unroll
        for (int c = 0; c < CPG; ++c) {
            if (c0 + c < CT) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const W4 xq = *reinterpret_cast<const W4*>(xb + ((c0 + c) * 2 + q) * 1024 + lane * 16);
                    acc[c] = mfma_half<PREC>(yt[q], xq, acc[c]);
                }
            }
        }
        if (bias_own) {
#pragma unroll
            for (int q = 0; q < 2; ++q) accb = mfma_half<PREC>(yt[q], ones, accb);
        }
#else
        acc[0] = mfma_half<PREC>(yt[0], yt[1], acc[0]);
#endif
    }
    float* out = p.partial + (((long)blockIdx.x * RT + rt) * NC) * 1024 + lane * 16;
    auto put = [&](int c, const f32x16& a) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = {a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
            *reinterpret_cast<f32x4*>(out + c * 1024 + 4 * q) = v;
        }
    };
#pragma unroll
    for (int c = 0; c < CPG; ++c)
        if (c0 + c < CT) put(c0 + c, acc[c]);
    if (bias_own) put(CT, accb);
}

// ------------------------------------------------------------------------------------------------
// wgrad(l) WITH dgrad(l) (round 4; asked for since round 1).  Both read the layer's gradient fragments G = d loss / d pre-activation
// [OUT = 256 channels x 32 samples] of every tile: k_wgrad to add G X^T to dW, k_dgrad_layer to form d X = W^T G.  Here the wgrad
// workgroup (8 wavefronts, wavefront w = row tile w of dW, RT = 8) also produces d X: after the tile's barrier all 16 gradient
// fragments lie in the LDS-DMA ring in exactly the form the matrix core takes as its B operand; wavefront w (< TO) multiplies them by
// the 16 A fragments of rows 32 w .. 32 w + 31 of W^T, which it keeps IN REGISTERS for its whole life (64 registers: no weight traffic
// at all), applies the ReLU bit mask of the layer below and stores its two fragments of d X.  Per tile the pair moved 65 KiB
// (dgrad: G in, mask, d X out; wgrad: G, X in); fused it is 49 KiB.  A second barrier per tile orders the ring: the DMA issued at the
// top of iteration i overwrites the slot every wavefront read G from in iteration i - 1.
struct WgradFusedParams {
    WgradParams w;
    const char* wt;         // W^T fragment stream of the layer (tile-major, 16 k-steps per tile: pack.h layer_transposed)
    char* out_store;        // the store again, writable
    int mask_slot, out_slot;
    int y_last_slot = -1;   // RT < 8: the slot of the LAST row tile's fragment pair when it does not follow the others (-1: it does)
    const char* ygen_wt = nullptr;      // YGEN: W^T stream of the 16-row layer ABOVE (8 output tiles x 1 k-step), see k_wgrad_dgrad
    // optional: the first rows_tiles output tiles of d X leave as float32 rows [nsamp, rows_stride] (loss scale removed) instead of fragments
    float* rows = nullptr;
    int rows_stride = 0, rows_tiles = 0;
    long nsamp = 0;
    const unsigned* maxbits = nullptr;
};

constexpr int FUSED_WL = 6;        // W^T fragments per wavefront kept in LDS instead of registers (k_wgrad_dgrad)
#ifdef EVD_WD_STAMP
__device__ float g_wd_stamp[32 * 2048 * 8];
#endif

// RT_ < 8 (round 5: sigma_net.1 = [geo rows | sigma row], 4 + 1 row tiles): wavefronts RT_ .. 7 own no gradient rows (no wgrad products,
// no partial blocks), the dgrad runs over the first KD_ gradient fragments (the last row tile may be a single fragment: KD_ = 2 RT_ - 1).
// YGEN (round 5: color_net.1 under the 3-row color_net.2): the layer's gradient G is not read but FORMED -- every wavefront fetches the pair
// [gradient fragment of the layer above (16 rows) | ReLU bit-mask fragment of this layer's output] at y_slot (2 KiB, the same for all
// eight: L2 hits) in place of its 2 KiB of G, multiplies its 32 rows of the upper layer's W^T (one resident A fragment) with it, masks,
// and puts the two fragments where the DMA would have put them.  The upper layer's dgrad launch, its 16 KiB write and this kernel's
// 16 KiB read of G per tile are gone.
// ROWS: the first rows_tiles output tiles of d X leave as float32 rows (WgradFusedParams::rows) instead of fragments.
template <int PREC, int CT, int TO, int OMASK, int RT_ = 8, int KD_ = 16, bool YGEN = false, bool ROWS = false>
__global__ __launch_bounds__(WGRAD_NT) void k_wgrad_dgrad(const WgradFusedParams fp) {
    static_assert(!YGEN || (RT_ == 8 && KD_ == 16), "a formed gradient has all 8 row tiles");
    static_assert(is_half_prec(PREC) && CT <= 8 && TO <= 8 && (OMASK == 0 || TO <= CT), "half-precision fragments, 8 x 32 gradient rows; a masked d X tile is a column tile of X");
    static_assert(RT_ <= 8 && KD_ <= 2 * RT_ && KD_ > 2 * (RT_ - 1), "k-steps of the dgrad = the gradient fragments of the RT_ row tiles");
    constexpr int RT = 8, CPG = wgrad_cpg(RT, CT), KD = KD_, WL = KD_ > FUSED_WL + 2 ? FUSED_WL : 0;
    typedef POps<PREC> O;
    const WgradParams& p = fp.w;
    extern __shared__ __attribute__((aligned(16))) char wsm[];
    char* raw = wsm;                                          // [WG_RING][8 wavefronts][4 fragments][1 KiB]
    char* xs = wsm + WG_RING * 8 * 4 * 1024;                  // [CT][2][1 KiB] transposed activation blocks (single: two barriers per tile)
    char* wl = xs + CT * 2048;                                // [8 wavefronts][WL][1 KiB]: the last WL of a wavefront's 16 W^T fragments
    pipe_fp16_saturate<PREC>();
    const int NC = CT + (p.bias ? 1 : 0);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n = lane & 31, h = lane >> 5;
    const int rt = wave, c0 = 0;
    // (activation block w belongs to wavefront w.  Also measured, round 5: with no mask the LAST CT wavefronts taking the blocks, so that in
    // the 8 x 4 / 8 x 5 launches the dgrad wavefronts do not transpose as well -- same box 13.55 / 13.56 / 13.56 against 13.52 / 13.58 / 13.56 ms
    // per iteration: the wait moves between the two barriers, removed)
    const int xw = wave;
    const bool xown = xw >= 0 && xw < CT, yown = wave < RT_, bias_own = p.bias != 0 && yown, down = wave < TO;
    const unsigned one = half_one_pair<PREC>();
    f32x16 acc[CPG], accb;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        accb[i] = 0.f;
#pragma unroll
        for (int c = 0; c < CPG; ++c) acc[c][i] = 0.f;
    }
    // rows 32 w .. of W^T: 16 A fragments, resident -- KD - WL in registers, the last WL in this wavefront's own LDS slots (the kernel
    // needs 128 + 16 accumulators: with all 64 registers of W^T it spilled into its loop, and a scratch reload drains the VM counter)
    W4 wt[KD - WL];
#pragma unroll
    for (int j = 0; j < KD; ++j) {
        const W4 f = *reinterpret_cast<const W4*>(fp.wt + ((long)(down ? wave : 0) * KD + j) * 1024 + lane * 16);    // (wavefronts beyond TO: tile 0 again, unused)
        if (j < KD - WL) wt[j] = f;
        else *reinterpret_cast<W4*>(wl + (wave * WL + (j - (KD - WL))) * 1024 + lane * 16) = f;
    }
    // the tile's four fragments of this wavefront: y0, y1 (adjacent: one address, immediates 0 / 1024) and x0, x1 (wavefronts beyond CT:
    // the y pair again) -- two address registers (the instruction's immediate moves the source and the LDS destination together)
    // (wavefronts beyond RT_: row tile 0's pair again, unused -- every tile costs every wavefront the same four DMA instructions)
    const int ys = (YGEN || !yown) ? p.y_slot : (RT_ < 8 && rt == RT_ - 1 && fp.y_last_slot >= 0) ? fp.y_last_slot : p.y_slot + 2 * rt;
    const long oy = (long)ys * 1024 + lane * 16;
    const long ox = (xown ? (long)(p.x_slot + 2 * xw) * 1024 : (long)ys * 1024) + lane * 16 - 2048;
    const unsigned ring0 = lds_offset_of(raw) + wave * 4096;
    auto issue = [&](long t, int stage) {
        if (t >= p.tiles) return;
        const char* g = p.store + t * p.tile_bytes;
        const unsigned dst = __builtin_amdgcn_readfirstlane(ring0 + stage * (8 * 4096));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                     "global_load_lds_dwordx4 %2, off offset:2048\n\tglobal_load_lds_dwordx4 %2, off offset:3072\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g + oy), "v"(g + ox), "s"(dst) : "memory");
    };
    W4 w2 = {{0u, 0u, 0u, 0u}};
    if constexpr (YGEN) w2 = *reinterpret_cast<const W4*>(fp.ygen_wt + (long)wave * 1024 + lane * 16);       // rows 32 w .. of the upper layer's W^T
    const long stride = gridDim.x;
    long t = blockIdx.x;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the weight fragments have landed: the counted waits below count the ring's DMA only
    if constexpr (YGEN) asm volatile("" : "+v"(w2.w[0]), "+v"(w2.w[1]), "+v"(w2.w[2]), "+v"(w2.w[3]));
#pragma unroll
    for (int j = 0; j < KD - WL; ++j) asm volatile("" : "+v"(wt[j].w[0]), "+v"(wt[j].w[1]), "+v"(wt[j].w[2]), "+v"(wt[j].w[3]));
#pragma unroll
    for (int k = 0; k < WG_RING - 1; ++k) issue(t + k * stride, k);
#ifdef EVD_WD_STAMP     // developer build (tools/dev/stamp_wgrad_dgrad.sh): shader-clock cycles of this wavefront's phases summed over its tiles
    long long wph[6] = {0, 0, 0, 0, 0, 0}, wq0, wq1;
    int wtiles = 0;
#define EVD_WD_T0() wq0 = __builtin_readcyclecounter()
#define EVD_WD_T(i) { wq1 = __builtin_readcyclecounter(); wph[i] += wq1 - wq0; wq0 = wq1; }
#else
#define EVD_WD_T0()
#define EVD_WD_T(i)
#endif
    for (int it = 0; t < p.tiles; t += stride, ++it) {
        EVD_WD_T0();
        issue(t + (WG_RING - 1) * stride, (it + WG_RING - 1) % WG_RING);
        const int later = (t + stride < p.tiles ? 1 : 0) + (t + 2 * stride < p.tiles ? 1 : 0);
        if (later == 2) wait_vmcnt<8>();
        else if (later == 1) wait_vmcnt<4>();
        else wait_vmcnt<0>();
        EVD_WD_T(0);
        // the transposition selectors and the bias column are re-made per tile (a dozen VALU instructions) instead of living in 12 registers
        // next to 128 + 16 accumulators and the 64 registers of W^T: the kernel spilled into its loop, and scratch loads drain the VM counter
        W4 sel0, sel1, ones;
        {
            int nn = n, hh = h;
            asm volatile("" : "+v"(nn), "+v"(hh));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int kk = 8 * hh + 2 * e;
                sel0.w[e] = (nn == kk ? (one & 0xffffu) : 0u) | (nn == kk + 1 ? (one & 0xffff0000u) : 0u);
                sel1.w[e] = (nn == 16 + kk ? (one & 0xffffu) : 0u) | (nn == 17 + kk ? (one & 0xffff0000u) : 0u);
                ones.w[e] = nn == 0 ? one : 0u;
            }
        }
        char* rw = raw + ((it % WG_RING) * 8 + wave) * 4096 + lane * 16;
        W4 y0 = *reinterpret_cast<const W4*>(rw), y1 = *reinterpret_cast<const W4*>(rw + 1024);
        if constexpr (YGEN) {           // (y0, y1) = (gradient fragment of the layer above, this layer's ReLU bits) -> this wavefront's two fragments of G
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const f32x16 d = mfma_half<PREC>(w2, y0, zero);
            // byte j of a lane's 16 mask bytes = the ReLU bits of fragment j (mlp_pipe.h mask_from_bits): fragments 2 w, 2 w + 1 are one half
            // of word w / 2 -- read as ONE dword (indexing the four words by the wavefront number would go through scratch)
            const unsigned mw = *reinterpret_cast<const unsigned*>(rw + 1024 + 4 * (wave >> 1)) >> (16 * (wave & 1));
            typename O::B g2[2];
#pragma unroll
            for (int k = 0; k < 8; ++k) O::template set_pair<false>(g2[k >> 2], k & 3, d[2 * k], d[2 * k + 1]);
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned two = (mw >> (8 * f + 2 * e)) & 3u;
                    g2[f].w[e] &= ((0u - (two & 1u)) & 0xffffu) | ((0u - (two >> 1)) & 0xffff0000u);
                }
            y0 = __builtin_bit_cast(W4, g2[0]);
            y1 = __builtin_bit_cast(W4, g2[1]);
            *reinterpret_cast<W4*>(rw) = y0;                 // where the dgrad of all eight wavefronts reads G (behind the barrier below)
            *reinterpret_cast<W4*>(rw + 1024) = y1;
        }
        if (KD_ == 2 * RT_ - 1 && rt == RT_ - 1) y1 = W4{{0u, 0u, 0u, 0u}};       // the last row tile is ONE fragment: what lies behind it in the store is not this layer's (0 x inf = nan)
        W4 yt[2];
        if (RT_ == 8 || yown) transpose_block<PREC>(y0, y1, sel0, sel1, yt);
        char* xb = xs;
        if (xown) {
            const W4 x0 = *reinterpret_cast<const W4*>(rw + 2048), x1 = *reinterpret_cast<const W4*>(rw + 3072);
            W4 xt[2];
            transpose_block<PREC>(x0, x1, sel0, sel1, xt);
#pragma unroll
            for (int q = 0; q < 2; ++q) *reinterpret_cast<W4*>(xb + (xw * 2 + q) * 1024 + lane * 16) = xt[q];
        }
        EVD_WD_T(1);
        __syncthreads();
        EVD_WD_T(2);
#pragma unroll
        for (int c = 0; c < CPG; ++c) {
            if (c0 + c < CT && (RT_ == 8 || yown)) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const W4 xq = *reinterpret_cast<const W4*>(xb + ((c0 + c) * 2 + q) * 1024 + lane * 16);
                    acc[c] = mfma_half<PREC>(yt[q], xq, acc[c]);
                }
            }
        }
        if (bias_own) {
#pragma unroll
            for (int q = 0; q < 2; ++q) accb = mfma_half<PREC>(yt[q], ones, accb);
        }
        EVD_WD_T(3);
        // d X tile `wave` = W^T rows . G: gradient fragment j sits in wavefront j / 2's ring slot, fragment j & 1
        if (down) {
            f32x16 d = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const char* gb = raw + ((it % WG_RING) * 8) * 4096 + lane * 16;
#pragma unroll
            for (int j = 0; j < KD; ++j) {
                const W4 gj = *reinterpret_cast<const W4*>(gb + (j >> 1) * 4096 + (j & 1) * 1024);
                const W4 wj = j < KD - WL ? wt[j] : *reinterpret_cast<const W4*>(wl + (wave * WL + (j - (KD - WL))) * 1024 + lane * 16);
                d = mfma_half<PREC>(wj, gj, d);
            }
            typename O::B o2[2];
#pragma unroll
            for (int k = 0; k < 8; ++k) O::template set_pair<false>(o2[k >> 2], k & 3, d[2 * k], d[2 * k + 1]);
            char* dst = fp.out_store + t * p.tile_bytes + (long)(fp.out_slot + 2 * wave) * 1024 + lane * 16;
            // The ReLU pattern of the layer below = [its stored activation != 0], and the activation fragments 2 w, 2 w + 1 are the X operand
            // this wavefront fetched for its column tile: still in its ring slot.  (The separate dgrad kernel reads the bit-mask fragment; a
            // load of it here -- tracked by the compiler or not -- either drains the ring's prefetch or races with its own wait.)
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                if (OMASK != 0) {
                    const typename O::B xa = __builtin_bit_cast(typename O::B, *reinterpret_cast<const W4*>(rw + 2048 + f * 1024));
                    O::mask_act(o2[f], xa);
                }
                // (plain stores: the next launch reads these d X fragments at once; non-temporal measured 10.91 vs 10.89 ms per iteration, round 6)
                if (!(ROWS && wave < fp.rows_tiles)) *reinterpret_cast<f32x4*>(dst + f * 1024) = __builtin_bit_cast(f32x4, o2[f]);
            }
            if (ROWS && wave < fp.rows_tiles) {
                const long smp = t * 32 + n;
                if (smp < fp.nsamp)
                    frag_pair_to_row<PREC>(__builtin_bit_cast(W4, o2[0]), __builtin_bit_cast(W4, o2[1]), fp.rows + smp * fp.rows_stride + 32 * wave, h,
                                           grad_scale(*fp.maxbits, true));
            }
        }
        EVD_WD_T(4);
        __syncthreads();            // every wavefront is done with this tile's ring slots before any of them issues into the oldest one
        EVD_WD_T(5);
#ifdef EVD_WD_STAMP
        ++wtiles;
#endif
    }
#ifdef EVD_WD_STAMP
    if (lane == 0) {                // [kind = CT + 8 YGEN ...][block][wavefront][8] floats, read back by evd_debug_wd_stamps
        float* sp = g_wd_stamp + ((long)(CT + (YGEN ? 8 : 0) + (RT_ < 8 ? 16 : 0)) * 2048 + (long)blockIdx.x * 8 + wave) * 8;
        for (int i = 0; i < 6; ++i) sp[i] = (float)wph[i];
        sp[6] = (float)wtiles; sp[7] = -7.f;
    }
#endif
    if (RT_ < 8 && !yown) return;           // (the reduce reads RT_ row tiles of every workgroup's 8-tile set: WreduceParams::part_stride)
    float* out = p.partial + (((long)blockIdx.x * RT + rt) * NC) * 1024 + lane * 16;
    auto put = [&](int c, const f32x16& a) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = {a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
            *reinterpret_cast<f32x4*>(out + c * 1024 + 4 * q) = v;
        }
    };
#pragma unroll
    for (int c = 0; c < CPG; ++c)
        if (c0 + c < CT) put(c0 + c, acc[c]);
    if (bias_own) put(CT, accb);
}

// ------------------------------------------------------------------------------------------------
// k_wgrad_dgrad_p (round 6): the same launch for a layer WITHOUT a bias gradient (the PDRF levels of every shipped config:
// rgb_add_bias off, sigma_net has none), software-pipelined by one tile with ONE barrier per tile.
// The stamps of k_wgrad_dgrad (profiles/r05_wgrad_dgrad_stamps.log) show a tile as: [transposes, exchange write: 19-27 %] barrier [products:
// MFMA-bound] [dgrad: MFMA-bound] barrier -- all eight wavefronts in the same phase at the same time, so the two wavefronts of a SIMD fight
// for the matrix pipe in the product phases and leave it idle in the others (the pipe is busy 28-39 %).  Here
//   * the transposed activation blocks are double-buffered and the DMA for tile t + 2 is issued BEHIND the tile's barrier: the barrier at the
//     top of tile t proves that every wavefront has left tile t - 1, which is what both the ring slot and the exchange buffer wait for -- the
//     second barrier is gone;
//   * phase A of tile t + 1 (gradient forming, both transposes, exchange write) runs INSIDE tile t, and the two halves of the workgroup take
//     it at different places: wavefronts 0-3 (one per SIMD) run products -> A -> dgrad, wavefronts 4-7 A -> products -> dgrad, so that a
//     SIMD's two wavefronts are in complementary phases (matrix pipe beside LDS / VALU work);
//   * no bias accumulator: its 16 registers hold two more W^T fragments (4 instead of 6 per wavefront in LDS: 32 KiB) and the second set of
//     transposed gradient fragments; LDS = 96 (ring) + 32 (exchange, CT = 8) + 32 = 160 KiB.
constexpr int FUSED_WL_P = 4;
template <int PREC, int CT, int TO, int OMASK, int RT_ = 8, int KD_ = 16, bool YGEN = false, bool ROWS = false>
__global__ __launch_bounds__(WGRAD_NT) void k_wgrad_dgrad_p(const WgradFusedParams fp) {
    static_assert(!YGEN || (RT_ == 8 && KD_ == 16), "a formed gradient has all 8 row tiles");
    static_assert(is_half_prec(PREC) && CT <= 8 && TO <= 8 && (OMASK == 0 || TO <= CT), "half-precision fragments, 8 x 32 gradient rows; a masked d X tile is a column tile of X");
    static_assert(RT_ <= 8 && KD_ <= 2 * RT_ && KD_ > 2 * (RT_ - 1), "k-steps of the dgrad = the gradient fragments of the RT_ row tiles");
    constexpr int RT = 8, CPG = wgrad_cpg(RT, CT), KD = KD_, WL = KD_ > FUSED_WL_P + 2 ? FUSED_WL_P : 0;
    typedef POps<PREC> O;
    const WgradParams& p = fp.w;
    extern __shared__ __attribute__((aligned(16))) char wsm[];
    char* raw = wsm;                                          // [WG_RING][8 wavefronts][4 fragments][1 KiB]
    char* xs = wsm + WG_RING * 8 * 4 * 1024;                  // [2][CT][2][1 KiB] transposed activation blocks, double-buffered
    char* wl = xs + 2 * CT * 2048;                            // [8 wavefronts][WL][1 KiB]
    pipe_fp16_saturate<PREC>();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n = lane & 31, h = lane >> 5;
    const int rt = wave, xw = wave;
    const bool xown = xw < CT, yown = wave < RT_, down = wave < TO;
    const unsigned one = half_one_pair<PREC>();
    f32x16 acc[CPG];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int c = 0; c < CPG; ++c) acc[c][i] = 0.f;
    W4 wt[KD - WL];
#pragma unroll
    for (int j = 0; j < KD; ++j) {
        const W4 f = *reinterpret_cast<const W4*>(fp.wt + ((long)(down ? wave : 0) * KD + j) * 1024 + lane * 16);
        if (j < KD - WL) wt[j] = f;
        else *reinterpret_cast<W4*>(wl + (wave * WL + (j - (KD - WL))) * 1024 + lane * 16) = f;
    }
    const int ys = (YGEN || !yown) ? p.y_slot : (RT_ < 8 && rt == RT_ - 1 && fp.y_last_slot >= 0) ? fp.y_last_slot : p.y_slot + 2 * rt;
    const long oy = (long)ys * 1024 + lane * 16;
    const long ox = (xown ? (long)(p.x_slot + 2 * xw) * 1024 : (long)ys * 1024) + lane * 16 - 2048;
    const unsigned ring0 = lds_offset_of(raw) + wave * 4096;
    auto issue = [&](long t, int stage) {
        if (t >= p.tiles) return;
        const char* g = p.store + t * p.tile_bytes;
        const unsigned dst = __builtin_amdgcn_readfirstlane(ring0 + stage * (8 * 4096));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                     "global_load_lds_dwordx4 %2, off offset:2048\n\tglobal_load_lds_dwordx4 %2, off offset:3072\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g + oy), "v"(g + ox), "s"(dst) : "memory");
    };
    W4 w2 = {{0u, 0u, 0u, 0u}};
    if constexpr (YGEN) w2 = *reinterpret_cast<const W4*>(fp.ygen_wt + (long)wave * 1024 + lane * 16);
    const long stride = gridDim.x;
    long t = blockIdx.x;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (YGEN) asm volatile("" : "+v"(w2.w[0]), "+v"(w2.w[1]), "+v"(w2.w[2]), "+v"(w2.w[3]));
#pragma unroll
    for (int j = 0; j < KD - WL; ++j) asm volatile("" : "+v"(wt[j].w[0]), "+v"(wt[j].w[1]), "+v"(wt[j].w[2]), "+v"(wt[j].w[3]));

    // phase A of the tile in ring slot `slot`, exchange buffer `buf`: this wavefront's gradient pair (formed, if YGEN) transposed into yt_out,
    // its activation pair transposed into the exchange buffer
    auto phase_a = [&](int slot, int buf, W4 (&yt_out)[2]) {
        W4 sel0, sel1;
        {
            int nn = n, hh = h;
            asm volatile("" : "+v"(nn), "+v"(hh));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int kk = 8 * hh + 2 * e;
                sel0.w[e] = (nn == kk ? (one & 0xffffu) : 0u) | (nn == kk + 1 ? (one & 0xffff0000u) : 0u);
                sel1.w[e] = (nn == 16 + kk ? (one & 0xffffu) : 0u) | (nn == 17 + kk ? (one & 0xffff0000u) : 0u);
            }
        }
        char* rw = raw + (slot * 8 + wave) * 4096 + lane * 16;
        W4 y0 = *reinterpret_cast<const W4*>(rw), y1 = *reinterpret_cast<const W4*>(rw + 1024);
        if constexpr (YGEN) {
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const f32x16 d = mfma_half<PREC>(w2, y0, zero);
            const unsigned mw = *reinterpret_cast<const unsigned*>(rw + 1024 + 4 * (wave >> 1)) >> (16 * (wave & 1));
            typename O::B g2[2];
#pragma unroll
            for (int k = 0; k < 8; ++k) O::template set_pair<false>(g2[k >> 2], k & 3, d[2 * k], d[2 * k + 1]);
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned two = (mw >> (8 * f + 2 * e)) & 3u;
                    g2[f].w[e] &= ((0u - (two & 1u)) & 0xffffu) | ((0u - (two >> 1)) & 0xffff0000u);
                }
            y0 = __builtin_bit_cast(W4, g2[0]);
            y1 = __builtin_bit_cast(W4, g2[1]);
            *reinterpret_cast<W4*>(rw) = y0;                 // where the dgrad of all eight wavefronts reads G (behind the next barrier)
            *reinterpret_cast<W4*>(rw + 1024) = y1;
        }
        if (KD_ == 2 * RT_ - 1 && rt == RT_ - 1) y1 = W4{{0u, 0u, 0u, 0u}};
        if (RT_ == 8 || yown) transpose_block<PREC>(y0, y1, sel0, sel1, yt_out);
        if (xown) {
            const W4 x0 = *reinterpret_cast<const W4*>(rw + 2048), x1 = *reinterpret_cast<const W4*>(rw + 3072);
            W4 xt[2];
            transpose_block<PREC>(x0, x1, sel0, sel1, xt);
            char* xb = xs + buf * (CT * 2048);
#pragma unroll
            for (int q = 0; q < 2; ++q) *reinterpret_cast<W4*>(xb + (xw * 2 + q) * 1024 + lane * 16) = xt[q];
        }
    };
    auto products = [&](int buf, const W4 (&yt)[2]) {
        const char* xb = xs + buf * (CT * 2048);
#pragma unroll
        for (int c = 0; c < CPG; ++c) {
            if (c < CT && (RT_ == 8 || yown)) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const W4 xq = *reinterpret_cast<const W4*>(xb + (c * 2 + q) * 1024 + lane * 16);
                    acc[c] = mfma_half<PREC>(yt[q], xq, acc[c]);
                }
            }
        }
    };
    auto dgrad = [&](int slot, long tt) {
        if (!down) return;
        f32x16 d = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const char* gb = raw + (slot * 8) * 4096 + lane * 16;
        const char* rw = raw + (slot * 8 + wave) * 4096 + lane * 16;
#pragma unroll
        for (int j = 0; j < KD; ++j) {
            const W4 gj = *reinterpret_cast<const W4*>(gb + (j >> 1) * 4096 + (j & 1) * 1024);
            const W4 wj = j < KD - WL ? wt[j] : *reinterpret_cast<const W4*>(wl + (wave * WL + (j - (KD - WL))) * 1024 + lane * 16);
            d = mfma_half<PREC>(wj, gj, d);
        }
        typename O::B o2[2];
#pragma unroll
        for (int k = 0; k < 8; ++k) O::template set_pair<false>(o2[k >> 2], k & 3, d[2 * k], d[2 * k + 1]);
        char* dst = fp.out_store + tt * p.tile_bytes + (long)(fp.out_slot + 2 * wave) * 1024 + lane * 16;
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            if (OMASK != 0) {
                const typename O::B xa = __builtin_bit_cast(typename O::B, *reinterpret_cast<const W4*>(rw + 2048 + f * 1024));
                O::mask_act(o2[f], xa);
            }
            if (!(ROWS && wave < fp.rows_tiles)) *reinterpret_cast<f32x4*>(dst + f * 1024) = __builtin_bit_cast(f32x4, o2[f]);
        }
        if (ROWS && wave < fp.rows_tiles) {
            const long smp = tt * 32 + n;
            if (smp < fp.nsamp)
                frag_pair_to_row<PREC>(__builtin_bit_cast(W4, o2[0]), __builtin_bit_cast(W4, o2[1]), fp.rows + smp * fp.rows_stride + 32 * wave, h,
                                       grad_scale(*fp.maxbits, true));
        }
    };

    // prologue: tiles 0 and 1 in flight, phase A of tile 0
    issue(t, 0);
    issue(t + stride, 1);
    W4 yt[2] = {{{0u, 0u, 0u, 0u}}, {{0u, 0u, 0u, 0u}}};
    if (t < p.tiles) {
        if (t + stride < p.tiles) wait_vmcnt<4>(); else wait_vmcnt<0>();
        phase_a(0, 0, yt);
    }
    for (int it = 0; t < p.tiles; t += stride, ++it) {
        __syncthreads();                // exchange buffer it & 1 and (YGEN) the formed gradient of tile `it` are complete; every wavefront has left tile it - 1
        issue(t + 2 * stride, (it + 2) % WG_RING);
        const bool has_next = t + stride < p.tiles;
        const int slot = it % WG_RING, nslot = (it + 1) % WG_RING, buf = it & 1;
        W4 ytn[2] = {{{0u, 0u, 0u, 0u}}, {{0u, 0u, 0u, 0u}}};
        // the DMA of tile it + 1 has landed when at most the four pieces of tile it + 2 (issued above) are outstanding; the stores of tile
        // it - 1 are older than those and complete with it
        // (each phase's code exists ONCE, in a two-trip loop whose trips a wavefront takes in its own order: with the two orders written out
        // hipcc spilled 130-150 registers beside the 128 accumulators + 48 registers of W^T)
#pragma unroll 1
        for (int ph = 0; ph < 2; ++ph) {
            const bool do_products = (wave < 4) == (ph == 0);
            if (do_products) {
                products(buf, yt);
            } else if (has_next) {
                if (t + 2 * stride < p.tiles) wait_vmcnt<4>(); else wait_vmcnt<0>();
                phase_a(nslot, buf ^ 1, ytn);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        dgrad(slot, t);
        __builtin_amdgcn_sched_barrier(0);
        yt[0] = ytn[0];
        yt[1] = ytn[1];
    }
    if (RT_ < 8 && !yown) return;
    float* out = p.partial + (((long)blockIdx.x * RT + rt) * CT) * 1024 + lane * 16;
#pragma unroll
    for (int c = 0; c < CPG; ++c)
        if (c < CT) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = {acc[c][4 * q], acc[c][4 * q + 1], acc[c][4 * q + 2], acc[c][4 * q + 3]};
                *reinterpret_cast<f32x4*>(out + c * 1024 + 4 * q) = v;
            }
        }
}

template <int PREC, int CT, int TO, int OMASK, int RT_ = 8, int KD_ = 16, bool YGEN = false, bool ROWS = false>
static int launch_wgrad_dgrad(const WgradFusedParams& p, int blocks, hipStream_t st) {
    if (ROWS != (p.rows != nullptr)) return fail(EVD_E_INVALID, "k_wgrad_dgrad: the row output goes with the ROWS instantiation");
    // a layer without a bias gradient: the pipelined one-barrier form (EVD_BWD_PIPE=0: the two-barrier kernel for every layer)
    static const bool pipe_on = [] { const char* e = getenv("EVD_BWD_PIPE"); return !(e && e[0] == '0'); }();
    if (pipe_on && !p.w.bias && CT <= 5) {
        const size_t ldsp = (size_t)WG_RING * 8 * 4 * 1024 + (size_t)2 * CT * 2048 + (size_t)8 * FUSED_WL_P * 1024;
        if constexpr (CT <= 5) {
            EVD_SET_MAX_LDS((&k_wgrad_dgrad_p<PREC, CT, TO, OMASK, RT_, KD_, YGEN, ROWS>), ldsp);
            hipLaunchKernelGGL((k_wgrad_dgrad_p<PREC, CT, TO, OMASK, RT_, KD_, YGEN, ROWS>), dim3(blocks), dim3(WGRAD_NT), ldsp, st, p);
            EVD_LAUNCH_CHECK();
            return EVD_OK;
        }
    }
    const size_t lds = (size_t)WG_RING * 8 * 4 * 1024 + (size_t)CT * 2048 + (size_t)8 * FUSED_WL * 1024;
    EVD_SET_MAX_LDS((&k_wgrad_dgrad<PREC, CT, TO, OMASK, RT_, KD_, YGEN, ROWS>), lds);
    hipLaunchKernelGGL((k_wgrad_dgrad<PREC, CT, TO, OMASK, RT_, KD_, YGEN, ROWS>), dim3(blocks), dim3(WGRAD_NT), lds, st, p);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

struct WreduceParams {
    const float* partial;
    int nparts, RT, NC, CT;            // NC = CT + 1 when the bias column rides along
    const int* rowmap;                 // [RT * 32] -> row of dW (or -1)
    const int* colmap;                 // [CT * 32] -> column of dW (or -1)
    float* dW;
    int ld;
    float* db;                         // or null
    const unsigned* maxbits;
    int accum = 0;                     // 1: add into dW / db (the caller's persistent gradient buffers) instead of overwriting them
    long part_stride = 0;              // floats between two workgroups' block sets (0: RT * NC * 1024, the sets of ONE layer back to back)
};

// A block sums 64 float4 columns of the partials (4 slices of the workgroup list, folded through LDS), then scatters
// the 256 sums: accumulator element (rt, c, lane, i) -> dW[rowmap[..]][colmap[..]]
static __device__ __forceinline__ void wgrad_reduce_block(const WreduceParams& p, int bx) {
    __shared__ f32x4 fold[4][64];
    const long per = (long)p.RT * p.NC * 1024, v4 = (long)bx * 64 + (threadIdx.x & 63);
    const long pstride = p.part_stride ? p.part_stride : per;
    const int slice = threadIdx.x >> 6;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (v4 * 4 < per) {
        const f32x4* src = reinterpret_cast<const f32x4*>(p.partial) + v4;
#pragma unroll 4
        for (int q = slice; q < p.nparts; q += 4) s += src[(long)q * (pstride / 4)];
    }
    fold[slice][threadIdx.x & 63] = s;
    __syncthreads();
    if (slice != 0 || v4 * 4 >= per) return;
    s = (fold[0][threadIdx.x] + fold[1][threadIdx.x]) + (fold[2][threadIdx.x] + fold[3][threadIdx.x]);
    const float inv = grad_scale(*p.maxbits, true);
    const long idx = v4 * 4;
    const int lane = (idx >> 4) & 63, c = (int)((idx >> 10) % p.NC), rt = (int)((idx >> 10) / p.NC), nc = lane & 31;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = (int)(idx & 15) + k, nr = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
        const int row = p.rowmap[rt * 32 + nr];
        if (row < 0) continue;
        if (c < p.CT) {
            const int col = p.colmap[c * 32 + nc];
            if (col >= 0) {            // (row, col) is hit by exactly one thread of one launch: a plain read-modify-write
                float* w = p.dW + (long)row * p.ld + col;
                *w = p.accum ? *w + s[k] * inv : s[k] * inv;
            }
        } else if (nc == 0 && p.db) {
            p.db[row] = p.accum ? p.db[row] + s[k] * inv : s[k] * inv;
        }
    }
}
static __global__ __launch_bounds__(256) void k_wgrad_reduce(const WreduceParams p) { wgrad_reduce_block(p, blockIdx.x); }

// several layers' reductions in one launch (blockIdx.y = layer; the fused backward of voxel_bwd_fused64.h leaves all its layers' blocks
// in one set per workgroup); a layer whose gradient is not wanted has RT = 0
constexpr int WREDUCE_MAX_JOBS = 5;
struct WreduceJobs { WreduceParams j[WREDUCE_MAX_JOBS]; };
static __global__ __launch_bounds__(256) void k_wgrad_reduce_jobs(const WreduceJobs jobs) { wgrad_reduce_block(jobs.j[blockIdx.y], blockIdx.x); }

// The split-float16 (float32-grade) form: both operands are (hi, lo) pairs; each half is transposed on its own (exact), the products
// are yh xh into acc and yh xl + yl xh into a second accumulator (scaled by 2^11, like the forward's cross terms).  Plain global loads
// (no DMA ring: this mode is for gradient fidelity, the half-precision kernel above is the fast one).  Blocks whose accumulators
// would not fit (RT x CT = 8 x 8) walk the sample tiles once per half of their columns.
template <int RT, int CT, bool YSINGLE>
__global__ __launch_bounds__(WGRAD_NT) void k_wgrad_split(const WgradParams p) {
    constexpr int CPG = wgrad_cpg(RT, CT), PASSES = CPG > 4 ? 2 : 1, CPP = CPG / PASSES, FB = 2048, PREC = EVD_PREC_F16;
    static_assert(8 % RT == 0 && CT <= 8 && (!YSINGLE || RT == 1) && CPG % PASSES == 0, "shape of the block");
    extern __shared__ __attribute__((aligned(16))) char wsm[];
    char* xs = wsm;                                           // [2][CT][hi q0, hi q1, lo q0, lo q1][1 KiB]
    const int NC = CT + (p.bias ? 1 : 0);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n = lane & 31, h = lane >> 5;
    const int rt = wave % RT, c0 = (wave / RT) * CPG;
    const bool xown = wave < CT, bias_own = p.bias && wave < RT;
    W4 sel0, sel1, ones;
    const unsigned one = half_one_pair<PREC>();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int kk = 8 * h + 2 * e;
        sel0.w[e] = (n == kk ? (one & 0xffffu) : 0u) | (n == kk + 1 ? (one & 0xffff0000u) : 0u);
        sel1.w[e] = (n == 16 + kk ? (one & 0xffffu) : 0u) | (n == 17 + kk ? (one & 0xffff0000u) : 0u);
        ones.w[e] = n == 0 ? one : 0u;
    }
    const W4 zf = {{0u, 0u, 0u, 0u}};
    float* out = p.partial + (((long)blockIdx.x * RT + rt) * NC) * 1024 + lane * 16;
    auto put = [&](int c, const f32x16& a, const f32x16& ax) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = fmaf(ax[4 * q + i], 4.8828125e-4f, a[4 * q + i]);
            *reinterpret_cast<f32x4*>(out + c * 1024 + 4 * q) = v;
        }
    };
    int it = 0;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        f32x16 acc[CPP], accx[CPP], accb, accbx;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            accb[i] = accbx[i] = 0.f;
#pragma unroll
            for (int c = 0; c < CPP; ++c) acc[c][i] = accx[c][i] = 0.f;
        }
        for (long t = blockIdx.x; t < p.tiles; t += gridDim.x, ++it) {
            const char* g = p.store + t * p.tile_bytes + lane * 16;
            const W4 y0h = frag_load<W4, FB>(g, p.y_slot + 2 * rt), y0l = frag_load<W4, FB>(g + 1024, p.y_slot + 2 * rt);
            const W4 y1h = YSINGLE ? zf : frag_load<W4, FB>(g, p.y_slot + 2 * rt + 1), y1l = YSINGLE ? zf : frag_load<W4, FB>(g + 1024, p.y_slot + 2 * rt + 1);
            W4 yth[2], ytl[2];
            transpose_block<PREC>(y0h, y1h, sel0, sel1, yth);
            transpose_block<PREC>(y0l, y1l, sel0, sel1, ytl);
            char* xb = xs + (it & 1) * (CT * 4096);
            if (xown) {
                const W4 x0h = frag_load<W4, FB>(g, p.x_slot + 2 * wave), x0l = frag_load<W4, FB>(g + 1024, p.x_slot + 2 * wave);
                const W4 x1h = frag_load<W4, FB>(g, p.x_slot + 2 * wave + 1), x1l = frag_load<W4, FB>(g + 1024, p.x_slot + 2 * wave + 1);
                W4 xth[2], xtl[2];
                transpose_block<PREC>(x0h, x1h, sel0, sel1, xth);
                transpose_block<PREC>(x0l, x1l, sel0, sel1, xtl);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    *reinterpret_cast<W4*>(xb + (wave * 4 + q) * 1024 + lane * 16) = xth[q];
                    *reinterpret_cast<W4*>(xb + (wave * 4 + 2 + q) * 1024 + lane * 16) = xtl[q];
                }
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < CPP; ++c) {
                const int col = c0 + ps * CPP + c;
                if (col < CT) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const W4 xh = *reinterpret_cast<const W4*>(xb + (col * 4 + q) * 1024 + lane * 16);
                        const W4 xl = *reinterpret_cast<const W4*>(xb + (col * 4 + 2 + q) * 1024 + lane * 16);
                        acc[c] = mfma_half<PREC>(yth[q], xh, acc[c]);
                        accx[c] = mfma_half<PREC>(yth[q], xl, accx[c]);
                        accx[c] = mfma_half<PREC>(ytl[q], xh, accx[c]);
                    }
                }
            }
            if (bias_own && ps == 0) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    accb = mfma_half<PREC>(yth[q], ones, accb);
                    accbx = mfma_half<PREC>(ytl[q], ones, accbx);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < CPP; ++c)
            if (c0 + ps * CPP + c < CT) put(c0 + ps * CPP + c, acc[c], accx[c]);
        if (bias_own && ps == 0) put(CT, accb, accbx);
    }
}

template <int PREC, int RT, int CT, bool YSINGLE>
static int launch_wgrad(const WgradParams& p, int blocks, hipStream_t st) {
    if constexpr (PREC == EVD_PREC_F16X3) {
        const size_t lds = (size_t)2 * CT * 4096;
        EVD_SET_MAX_LDS((&k_wgrad_split<RT, CT, YSINGLE>), lds);
        hipLaunchKernelGGL((k_wgrad_split<RT, CT, YSINGLE>), dim3(blocks), dim3(WGRAD_NT), lds, st, p);
    } else {
        const size_t lds = wgrad_lds_bytes(CT);
        EVD_SET_MAX_LDS((&k_wgrad<PREC, RT, CT, YSINGLE>), lds);
        hipLaunchKernelGGL((k_wgrad<PREC, RT, CT, YSINGLE>), dim3(blocks), dim3(WGRAD_NT), lds, st, p);
    }
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

// Gradient of a positional encoding (embedding.py:88-98) from its gradient fragments (KSN fragments at `slot`, the encoding's own
// arrangement: position q = 8 j + e holds, for q < 3 L, d sin (lane half 0) / d cos (half 1) of x_{q % 3} 2^{q / 3}; q = 3 L:
// (d x_0 | d x_1); q = 3 L + 1: (d x_2 | -)) -> d x rows [n, 3] float32, loss scale removed.  x: rows of x_stride floats,
// one per `per` samples (1: points; S: the ray's view direction).
template <int PREC, int L, int KSN>
__global__ __launch_bounds__(256) void k_pe_bwd(const char* __restrict__ store, long tile_bytes, int slot, long nsamp, const float* __restrict__ x,
                                                int x_stride, int per, const unsigned* __restrict__ maxbits, float* __restrict__ dx, int accumulate) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x, tile = idx >> 6;
    const int lane = idx & 63, n = lane & 31, h = lane >> 5;
    const long smp = tile * 32 + n;
    if (tile * 32 >= nsamp) return;
    const long sc = smp < nsamp ? smp : nsamp - 1;
    float xv[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) xv[c] = x[(sc / per) * x_stride + c];
    float g[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < KSN; ++j) {
        float fv[8];
        frag_values<PREC>(store + tile * tile_bytes + lane * 16, slot + j, fv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int q = 8 * j + e;
            const float v = fv[e];
            if (q < 3 * L) {
                const float fr = (float)(1 << (q / 3)), a = xv[q % 3] * fr;
                // d sin = cos, d cos = -sin.  The half-precision modes take them from the hardware unit, as their forward does (sin_or_cos_hw:
                // absolute error ~2^-20 against gradient fragments rounded at 2^-11; libm's cosf / sinf with their range reduction were 57 M VALU
                // instructions per launch -- profiles/r05_pmc_train_step.txt --, the float32-grade mode keeps them)
                const float tr = POps<PREC>::kFastTrig ? sin_or_cos_hw(a, h == 0 ? 1 : 0) : (h == 0 ? cosf(a) : sinf(a));
                g[q % 3] += h == 0 ? v * tr * fr : -v * tr * fr;
            } else if (q == 3 * L) {
                g[h] += v;                       // (x_0 | x_1)
            } else if (q == 3 * L + 1 && h == 0) {
                g[2] += v;
            }
        }
    }
    const float inv = grad_scale(*maxbits, true);
#pragma unroll
    for (int c = 0; c < 3; ++c) g[c] = (g[c] + __shfl_xor(g[c], 32)) * inv;
    if (h == 0 && smp < nsamp) {
#pragma unroll
        for (int c = 0; c < 3; ++c) dx[smp * 3 + c] = accumulate ? dx[smp * 3 + c] + g[c] : g[c];
    }
}

// ------------------------------------------------------------------------------------------------
// The whole backward of one network, in the order the gradients become available.
template <int PREC> static int run_nerf_backward(const BwdPlan& b, hipStream_t st) {
    using namespace astore;
    constexpr int D = 8;
    int rc;
    EVD_HIP(hipMemsetAsync(b.maxbits, 0, sizeof(unsigned), st));
    hipLaunchKernelGGL(k_absmax, dim3(512), dim3(256), 0, st, b.d_raw, b.nsamp * 4, b.maxbits);
    EVD_LAUNCH_CHECK();
    hipLaunchKernelGGL((k_grad_frags<PREC>), dim3((unsigned)cdiv(b.tiles * 64, 256L)), dim3(256), 0, st, b.d_raw, b.nsamp, b.maxbits, b.store, b.tiles);
    EVD_LAUNCH_CHECK();

    auto dgrad = [&](int stream, int in_slot, int extra_slot, int mask_slot, int out_slot) {
        DgradParams p;
        p.wstream = b.wt[stream]; p.store = b.store; p.tile_bytes = astore::tile_bytes(PREC); p.in_slot = in_slot; p.extra_slot = extra_slot; p.mask_slot = mask_slot; p.out_slot = out_slot;
        return p;
    };
    // wgrad + reduce of one parameter block: rows from `ymap`, columns from `xmap` (offset into b.maps)
    auto wgrad = [&](auto launch, int RT, int CT, bool bias, int y_slot, int x_slot, int ymap, int xmap, float* dW, int ld, float* db) -> int {
        if (!dW) return EVD_OK;
        const int blocks = (int)(cdiv(b.tiles, (long)WGRAD_TPI) < b.wgrad_blocks ? cdiv(b.tiles, (long)WGRAD_TPI) : b.wgrad_blocks);
        WgradParams p;
        p.store = b.store; p.tiles = b.tiles; p.tile_bytes = astore::tile_bytes(PREC); p.y_slot = y_slot; p.x_slot = x_slot; p.bias = bias ? 1 : 0; p.partial = b.partial;
        hipStream_t ws = st;
        if (b.side) {                           // everything issued so far on the main stream (the producer of this wgrad's operands) first
            EVD_HIP(hipEventRecord(b.ev, st));
            EVD_HIP(hipStreamWaitEvent(b.side, b.ev, 0));
            ws = b.side;
            if (int rcs = test_side_spin(ws)) return rcs;
        }
        int r = launch(p, blocks, ws);
        if (r) return r;
        WreduceParams q;
        q.partial = b.partial; q.nparts = blocks; q.RT = RT; q.CT = CT; q.NC = CT + (bias ? 1 : 0);
        q.rowmap = b.maps + ymap; q.colmap = b.maps + xmap; q.dW = dW; q.ld = ld; q.db = bias ? db : nullptr; q.maxbits = b.maxbits; q.accum = b.accumulate;
        hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((long)RT * q.NC * 4)), dim3(256), 0, ws, q);
        EVD_LAUNCH_CHECK();
        return EVD_OK;
    };
    // wgrad(l) with dgrad(l) in one launch (k_wgrad_dgrad above): the seven 256 x 256 trunk layers in the half-precision modes;
    // EVD_BWD_FUSE=0 keeps the separate launches (A/B)
    static const bool fuse_on = [] { const char* e = getenv("EVD_BWD_FUSE"); return !(e && e[0] == '0'); }();
    auto fused = [&](auto launch, int CT, bool bias, int y_slot, int x_slot, int ymap, int xmap, float* dW, int ld, float* db, int stream, int out_slot) -> int {
        const int blocks = (int)(b.tiles < b.wgrad_blocks ? b.tiles : b.wgrad_blocks);
        WgradFusedParams p;
        p.w.store = b.store; p.w.tiles = b.tiles; p.w.tile_bytes = astore::tile_bytes(PREC); p.w.y_slot = y_slot; p.w.x_slot = x_slot; p.w.bias = bias ? 1 : 0; p.w.partial = b.partial;
        p.wt = b.wt[stream]; p.out_store = b.store; p.mask_slot = -1; p.out_slot = out_slot;
        if (b.side && !test_skip_side_join()) { // wgrad launches in flight on the side stream use the partial scratch (and read what this one writes next): join
            EVD_HIP(hipEventRecord(b.ev, b.side));
            EVD_HIP(hipStreamWaitEvent(st, b.ev, 0));
        }
        int r = launch(p, blocks, st);
        if (r) return r;
        WreduceParams q;
        q.partial = b.partial; q.nparts = blocks; q.RT = 8; q.CT = CT; q.NC = CT + (bias ? 1 : 0);
        q.rowmap = b.maps + ymap; q.colmap = b.maps + xmap; q.dW = dW; q.ld = ld; q.db = bias ? db : nullptr; q.maxbits = b.maxbits; q.accum = b.accumulate;
        hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((long)8 * q.NC * 4)), dim3(256), 0, st, q);
        EVD_LAUNCH_CHECK();
        return EVD_OK;
    };
    const BwdGrads& g = b.grads;
    // Every wgrad is issued BEFORE the dgrad layer that reads the same two arrays (incoming gradient, saved activation): the two are
    // independent, and with a side stream (b.side: developer switch EVD_BWD_OVERLAP) they run concurrently and share those reads in
    // the Infinity Cache.
    // rgb_linear: d hv = Wr^T d rgb;  dWr = d rgb . hv^T
    if ((rc = wgrad(launch_wgrad<PREC, 1, 4, true>, 1, 4, true, G_RGB, HV, MAP_RGB, MAP_HID, g.rgb_w, 128, g.rgb_b))) return rc;
    if ((rc = launch_dgrad<PREC, 1, 4, 1, false, 2>(dgrad(EVD_BWD_RGB, G_RGB, -1, M_HV, D_HV), b.tiles, st))) return rc;
    // views_linears.0 on cat([feature, PE(dir)])
    if ((rc = wgrad(launch_wgrad<PREC, 4, 8, false>, 4, 8, true, D_HV, F, MAP_HID, MAP_HID, g.views_w, 256 + 27, g.views_b))) return rc;
    if ((rc = wgrad(launch_wgrad<PREC, 4, 1, false>, 4, 1, false, D_HV, DIR, MAP_HID, MAP_DIR, g.views_w, 256 + 27, nullptr))) return rc;
    if ((rc = launch_dgrad<PREC, 8, 8, 8, false, 0>(dgrad(EVD_BWD_VIEWS, D_HV, -1, -1, D_F), b.tiles, st))) return rc;
    // feature_linear and alpha_linear both read h_7
    if ((rc = wgrad(launch_wgrad<PREC, 8, 8, false>, 8, 8, true, D_F, H0 + 16 * (D - 1), MAP_HID, MAP_HID, g.feature_w, 256, g.feature_b))) return rc;
    if ((rc = wgrad(launch_wgrad<PREC, 1, 8, true>, 1, 8, true, G_ALPHA, H0 + 16 * (D - 1), MAP_ALPHA, MAP_HID, g.alpha_w, 256, g.alpha_b))) return rc;
    if ((rc = launch_dgrad<PREC, 17, 8, 16, true, 2>(dgrad(EVD_BWD_HEAD, D_F, G_ALPHA, M_H0 + D - 1, D_H0 + 16 * (D - 1)), b.tiles, st))) return rc;
    // pts_linears[l], l = 7 .. 1
    for (int l = D - 1; l >= 1; --l) {
        const bool wide = l - 1 == b.skip;
        if constexpr (is_half_prec(PREC)) {
            if (fuse_on && g.pts_w[l]) {
                if (wide && (rc = wgrad(launch_wgrad<PREC, 8, 2, false>, 8, 2, false, D_H0 + 16 * l, PE, MAP_HID, MAP_PE, g.pts_w[l], 256 + 63, nullptr))) return rc;
                if ((rc = fused(launch_wgrad_dgrad<PREC, 8, 8, 1>, 8, true, D_H0 + 16 * l, H0 + 16 * (l - 1), MAP_HID, wide ? MAP_HID_SKIP : MAP_HID, g.pts_w[l],
                                wide ? 256 + 63 : 256, g.pts_b[l], EVD_BWD_HIDDEN1 + l - 1, D_H0 + 16 * (l - 1)))) return rc;
                continue;
            }
        }
        if ((rc = wgrad(launch_wgrad<PREC, 8, 8, false>, 8, 8, true, D_H0 + 16 * l, H0 + 16 * (l - 1), MAP_HID, wide ? MAP_HID_SKIP : MAP_HID,
                        g.pts_w[l], wide ? 256 + 63 : 256, g.pts_b[l]))) return rc;
        if (wide && (rc = wgrad(launch_wgrad<PREC, 8, 2, false>, 8, 2, false, D_H0 + 16 * l, PE, MAP_HID, MAP_PE, g.pts_w[l], 256 + 63, nullptr))) return rc;
        if ((rc = launch_dgrad<PREC, 16, 8, 16, false, 2>(dgrad(EVD_BWD_HIDDEN1 + l - 1, D_H0 + 16 * l, -1, M_H0 + l - 1, D_H0 + 16 * (l - 1)), b.tiles, st))) return rc;
    }
    // gradient w.r.t. the rays: the encoding rows of pts_linears[0], of the skip layer and of views_linears.0, then through sin / cos
    if (b.d_pts) {
        if ((rc = launch_dgrad<PREC, 16, 2, 16, false, 0>(dgrad(EVD_BWD_PE0, D_H0, -1, -1, D_PE0), b.tiles, st))) return rc;
        hipLaunchKernelGGL((k_pe_bwd<PREC, PE_L, PE_KS>), dim3((unsigned)cdiv(b.tiles * 64, 256L)), dim3(256), 0, st, (const char*)b.store, astore::tile_bytes(PREC), D_PE0,
                           b.nsamp, b.pts, 3, 1, b.maxbits, b.d_pts, 0);
        EVD_LAUNCH_CHECK();
        if (b.skip >= 0 && b.skip + 1 < D) {
            if ((rc = launch_dgrad<PREC, 16, 2, 16, false, 0>(dgrad(EVD_BWD_PESKIP, D_H0 + 16 * (b.skip + 1), -1, -1, D_PE5), b.tiles, st))) return rc;
            hipLaunchKernelGGL((k_pe_bwd<PREC, PE_L, PE_KS>), dim3((unsigned)cdiv(b.tiles * 64, 256L)), dim3(256), 0, st, (const char*)b.store, astore::tile_bytes(PREC), D_PE5,
                               b.nsamp, b.pts, 3, 1, b.maxbits, b.d_pts, 1);
            EVD_LAUNCH_CHECK();
        }
    }
    if (b.d_dirs) {
        if ((rc = launch_dgrad<PREC, 8, 1, 8, false, 0>(dgrad(EVD_BWD_DIR, D_HV, -1, -1, D_DIRG), b.tiles, st))) return rc;
        hipLaunchKernelGGL((k_pe_bwd<PREC, PE_LV, PEV_KS>), dim3((unsigned)cdiv(b.tiles * 64, 256L)), dim3(256), 0, st, (const char*)b.store, astore::tile_bytes(PREC), D_DIRG,
                           b.nsamp, b.viewdirs, b.vd_stride, b.S, b.maxbits, b.d_dirs, 0);
        EVD_LAUNCH_CHECK();
    }
    // pts_linears[0] on PE(pts) (no dgrad beyond the inputs)
    if ((rc = wgrad(launch_wgrad<PREC, 8, 2, false>, 8, 2, true, D_H0, PE, MAP_HID, MAP_PE, g.pts_w[0], 63, g.pts_b[0]))) return rc;
    if (b.side && !test_skip_side_join()) {     // join the side stream
        EVD_HIP(hipEventRecord(b.ev, b.side));
        EVD_HIP(hipStreamWaitEvent(st, b.ev, 0));
    }
    return EVD_OK;
}

}  // namespace evd
