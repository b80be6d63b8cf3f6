// Device-side building blocks shared by the fused MLP kernels (NeRF backbone, PDRF sigma/colour nets):
// precision policies, the LDS weight stream, one transposed linear layer on a wavefront's 32 samples, and the
// branch-free sin/cos used by the in-register positional encoding.  Contract: nerf_mlp.h.
#pragma once

#include "evd_common.h"
#include "nerf_mlp.h"

namespace evd {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));


// ---------------------------------------------------------------------------------------------
// precision policies
template <int PREC> struct Ops;

template <> struct Ops<EVD_PREC_BF16> {
    typedef bf16x8 B;
    typedef bf16x8 A;
    static constexpr bool kSplit = false;
    static __device__ __forceinline__ A load_a(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }
    static __device__ __forceinline__ void mma(f32x16& acc, f32x16&, const A& a, const B& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    static __device__ __forceinline__ B make_b(const f32x8& v) { return __builtin_convertvector(v, bf16x8); }
};

template <> struct Ops<EVD_PREC_F16> {
    typedef f16x8 B;
    typedef f16x8 A;
    static constexpr bool kSplit = false;
    static __device__ __forceinline__ A load_a(const char* p) { return *reinterpret_cast<const f16x8*>(p); }
    static __device__ __forceinline__ void mma(f32x16& acc, f32x16&, const A& a, const B& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
    static __device__ __forceinline__ B make_b(const f32x8& v) {
        f32x8 c;
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = fminf(fmaxf(v[i], -65000.f), 65000.f);   // float16 range guard
        return __builtin_convertvector(c, f16x8);
    }
};

template <> struct Ops<EVD_PREC_F16X3> {
    struct B { f16x8 hi, lo; };
    struct A { f16x8 hi, lo; };
    static constexpr bool kSplit = true;
    static __device__ __forceinline__ A load_a(const char* p) {
        A a;
        a.hi = *reinterpret_cast<const f16x8*>(p);
        a.lo = *reinterpret_cast<const f16x8*>(p + 1024);
        return a;
    }
    static __device__ __forceinline__ void mma(f32x16& acc, f32x16& accx, const A& a, const B& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi, b.hi, acc, 0, 0, 0);
        accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi, b.lo, accx, 0, 0, 0);
        accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.lo, b.hi, accx, 0, 0, 0);
    }
    static __device__ __forceinline__ B make_b(const f32x8& v) {
        B b;
        f32x8 c;
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = fminf(fmaxf(v[i], -65000.f), 65000.f);   // float16 range guard
        b.hi = __builtin_convertvector(c, f16x8);
        const f32x8 back = __builtin_convertvector(b.hi, f32x8);
        b.lo = __builtin_convertvector((c - back) * 2048.f, f16x8);
        return b;
    }
};

template <> struct Ops<EVD_PREC_F32> {
    typedef f32x8 B;
    typedef f32x8 A;
    static constexpr bool kSplit = false;
    static __device__ __forceinline__ A load_a(const char* p) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(p), hi = *reinterpret_cast<const f32x4*>(p + 1024);
        return A{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
    static __device__ __forceinline__ void mma(f32x16& acc, f32x16&, const A& a, const B& b) {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
    }
    static __device__ __forceinline__ B make_b(const f32x8& v) { return v; }
};

// ---------------------------------------------------------------------------------------------
// Weight stream.  The packed fragments travel global -> LDS by LDS-DMA (global_load_lds_dwordx4: one wavefront
// instruction copies 64 x 16 B = one 1 KiB piece straight into LDS, no staging registers) into a ring of
// NSLOT chunks shared by all wavefronts of the workgroup.  Protocol, chunk c resident in slot c % 3:
//   chunk_begin(c): issue the DMA of chunk c+2 (its slot held chunk c-1, which every wavefront finished
//                   reading before the barrier that ended chunk c-1)
//   chunk_end(c):   counted s_waitcnt vmcnt(4) -- this wavefront's pieces of chunk c+1 have landed while the
//                   4 pieces of chunk c+2 stay in flight -- then ONE s_barrier.
// The DMA and the waits are inline asm, so hipcc neither drains them early nor orders them; the "memory"
// clobbers keep its LDS reads on the right side of the barrier.
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ unsigned lds_offset_of(const void* p) {
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char*)p;
}

template <int PREC> struct Stream {
    static constexpr int NT = mlp_threads(PREC);
    static constexpr int CB = chunk_bytes(PREC);
    static constexpr int FB = frag_bytes(PREC);
    static constexpr int FPC = frags_per_chunk(PREC);
    static constexpr int NSLOT = 3;
    static constexpr int PIECES = CB / 1024 / (NT / 64);   // 1 KiB DMA pieces per wavefront per chunk
    static_assert(PIECES == 4, "the counted vmcnt below assumes 4 pieces per wavefront per chunk");
    const char* gsrc;       // this lane's source address of piece 0 of chunk 0
    const char* lds;        // ring base (generic pointer, for the fragment reads)
    unsigned dst0;          // LDS byte offset of this wavefront's piece 0 in slot 0 (wave-uniform)
    int nchunks, cur, slot; // current chunk and its slot
    __device__ __forceinline__ void issue(int c, int sl) {
        if (c < nchunks) {
            const char* src = gsrc + (size_t)c * CB;
            const unsigned dst = dst0 + sl * CB;
#pragma unroll
            for (int t = 0; t < PIECES; ++t) dma16(src + t * 1024, dst + t * 1024);
        }
    }
    __device__ __forceinline__ void start(const char* g, char* ring, int n, int tid) {
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
        gsrc = g + wave * (PIECES * 1024) + lane * 16;
        lds = ring;
        dst0 = lds_offset_of(ring) + wave * (PIECES * 1024);
        nchunks = n; cur = 0; slot = 0;
        issue(0, 0);
        issue(1, 1);
        if (n > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // fragment fc of the resident chunk, this lane's 16 bytes
    __device__ __forceinline__ const char* frag(int fc, int lane) const { return lds + slot * CB + fc * FB + lane * 16; }
    __device__ __forceinline__ void chunk_begin() { issue(cur + 2, slot == 0 ? 2 : slot - 1); }   // (slot + 2) % 3
    __device__ __forceinline__ void chunk_end() {
        if (cur + 2 < nchunks) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifndef EVD_ABLATE_BARRIER
        __builtin_amdgcn_s_barrier();
#endif
        asm volatile("" ::: "memory");
        ++cur;
        slot = slot == 2 ? 0 : slot + 1;
    }
};

enum { OUT_B = 1, OUT_F32 = 2, OUT_BOTH = 3 };

// One linear layer on the wavefront's 32 samples.  in[KSTEPS] are B fragments; the output is the next layer's
// B fragments (OUT_B: out[2*TILES]) and/or the raw float32 D fragment of the (single) tile (OUT_F32).
// FOFF = fragment offset inside the current chunk at entry (static); the layer leaves the stream at
// (FOFF + TILES*KSTEPS) % FPC, or chunk-aligned when PAD_END.  bias points into LDS.
// A fragments are read PD fragments ahead of their MFMA (never across a chunk boundary: the next chunk is only
// guaranteed resident after the barrier), so LDS latency hides behind the previous MFMAs.
template <int PREC, int KSTEPS, int TILES, bool RELU, int OUT, int FOFF, bool PAD_END>
__device__ __forceinline__ void layer(Stream<PREC>& st, const typename Ops<PREC>::B (&in)[KSTEPS],
                                      typename Ops<PREC>::B* __restrict__ out, float* __restrict__ out_f32,
                                      const float* __restrict__ bias, int lane, float* __restrict__ feat_row, int W) {
    typedef Ops<PREC> O;
    typedef typename O::A A;
    constexpr int FPC = Stream<PREC>::FPC;
    constexpr int G = TILES >= 2 ? 2 : 1;
    constexpr int NF = TILES * KSTEPS;
#ifdef EVD_PD
    constexpr int PD = PREC == EVD_PREC_F32 ? 2 : EVD_PD;
#else
    constexpr int PD = PREC == EVD_PREC_F32 ? 2 : 4;        // prefetch depth in fragments
#endif
    static_assert(TILES % G == 0, "tile count must be a multiple of the accumulation group");
    const int h = lane >> 5;
    A abuf[PD];
    f32x16 acc[G], accx[G];
    // fragments [0, need(f)) have been requested once step f is reached
    auto need = [](int f) constexpr {
        int lim = f + PD;
        const int chunk_end_excl = ((FOFF + f) / FPC + 1) * FPC - FOFF;
        if (lim > chunk_end_excl) lim = chunk_end_excl;
        return lim > NF ? NF : lim;
    };
#pragma unroll
    for (int p = 0; p < TILES / G; ++p) {
#pragma unroll
        for (int j = 0; j < KSTEPS; ++j) {
#pragma unroll
            for (int t = 0; t < G; ++t) {
                const int f = (p * KSTEPS + j) * G + t;             // static after unrolling
                const int fc = (FOFF + f) % FPC;
                if (fc == 0) st.chunk_begin();
                const int lo = f == 0 ? 0 : need(f - 1), hi = need(f);
#pragma unroll
                for (int gq = 0; gq < PD; ++gq) {
                    const int gf = lo + gq;
#ifdef EVD_ABLATE_LDS
                    if (gf < hi && gf < PD) abuf[gf % PD] = O::load_a(st.frag((FOFF + gf) % FPC, lane));
#else
                    if (gf < hi) abuf[gf % PD] = O::load_a(st.frag((FOFF + gf) % FPC, lane));
#endif
                }
                if (j == 0) {        // accumulators start from the bias (C layout rows (r&3) + 8(r>>2) + 4h)
                    const float* bt = bias + (p * G + t) * 32 + 4 * h;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(bt + 8 * q);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { acc[t][4 * q + e] = bv[e]; accx[t][4 * q + e] = 0.f; }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);      // keep the prefetch reads ABOVE this MFMA (hipcc sinks them otherwise)
                O::mma(acc[t], accx[t], abuf[f % PD], in[j]);
                if (fc == FPC - 1) st.chunk_end();
            }
        }
#pragma unroll
        for (int t = 0; t < G; ++t) {
            const int tile = p * G + t;
            f32x8 v[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float x = O::kSplit ? fmaf(accx[t][r], 4.8828125e-4f, acc[t][r]) : acc[t][r];
                if (RELU) x = fmaxf(x, 0.f);
                v[r >> 3][r & 7] = x;
                if (OUT & OUT_F32) out_f32[r] = x;
            }
            if (feat_row) {          // float32 feature rows: features 32 tile + 8q + 4h + (0..3)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 w4 = {v[q >> 1][(q & 1) * 4], v[q >> 1][(q & 1) * 4 + 1], v[q >> 1][(q & 1) * 4 + 2], v[q >> 1][(q & 1) * 4 + 3]};
                    *reinterpret_cast<f32x4*>(feat_row + 32 * tile + 8 * q + 4 * h) = w4;
                }
            }
            if (OUT & OUT_B) {
#ifdef EVD_ABLATE_EPI
                out[2 * tile] = in[0];
                out[2 * tile + 1] = in[1];
                asm volatile("" :: "v"(v[0][0]), "v"(v[1][7]));
#else
                out[2 * tile] = O::make_b(v[0]);
                out[2 * tile + 1] = O::make_b(v[1]);
#endif
            }
        }
    }
    if (PAD_END && ((FOFF + NF) % FPC) != 0) st.chunk_end();
}

// LDS carve-up of the fused MLP kernels: [weight ring | biases | per-wavefront B-fragment stash]
template <int PREC> struct MlpLds {
    static constexpr int RING = Stream<PREC>::NSLOT * Stream<PREC>::CB;
    static constexpr int BIAS_FLOATS = 4096;                              // 16 KiB: enough for D <= 12 at W = 256
    static constexpr int STASH_PER_WAVE = PE_KS * 64 * (int)sizeof(typename Ops<PREC>::B);
    static constexpr int STASH = (mlp_threads(PREC) / 64) * STASH_PER_WAVE;
    static constexpr int TOTAL = RING + BIAS_FLOATS * 4 + STASH;
};

// cooperative copy of the bias block into LDS (before Stream::start, whose barrier publishes it)
template <int PREC>
__device__ __forceinline__ float* stage_bias(char* smem, const float* gbias, int nfloats, int tid) {
    float* b = reinterpret_cast<float*>(smem + MlpLds<PREC>::RING);
    for (int i = tid; i < nfloats; i += mlp_threads(PREC)) b[i] = gbias[i];
    return b;
}

// sin(a) for h == 0, cos(a) for h == 1, with one shared code path: Cody-Waite reduction to [-pi/4, pi/4]
// (3 fmaf terms, good for |a| < 1e5; beyond that the reduction runs in float64, accurate to ~1e-7 up to
// |a| ~ 1e9 -- positional-encoding arguments are 2^k x with x inside the scene box) and a quadrant-selected
// minimax polynomial (max error 7e-8 against float64 sin/cos on the fast path).
__device__ __forceinline__ float sin_or_cos(float a, int h) {
    float j, r;
    if (__builtin_expect(fabsf(a) <= 1.0e5f, 1)) {
        j = rintf(a * 0.636619772f);
        r = fmaf(-j, 1.57079601e+00f, a);
        r = fmaf(-j, 3.13916473e-07f, r);
        r = fmaf(-j, 5.39030253e-15f, r);
    } else {
        const double ad = (double)a;
        const double jd = rint(ad * 0.63661977236758134308);
        const double rd = fma(-jd, 6.123233995736766036e-17, fma(-jd, 1.57079632679489655800, ad));
        r = (float)rd;
        j = (float)fmod(jd, 4.0);
    }
    const int q = ((int)j + h) & 3;
    const float s = r * r;
    const bool use_cos = q & 1;
    // sin: r + r s (S1 + s (S2 + s (S3 + s S4)));   cos: 1 + s (C1 + s (C2 + s (C3 + s C4)))
    float p = use_cos ? 2.44677067e-5f : 2.86567956e-6f;
    p = fmaf(p, s, use_cos ? -1.38877297e-3f : -1.98559923e-4f);
    p = fmaf(p, s, use_cos ? 4.16666567e-2f : 8.33338592e-3f);
    p = fmaf(p, s, use_cos ? -0.5f : -1.66666672e-1f);
    const float base = use_cos ? 1.f : r;
    const float mul = use_cos ? s : s * r;
    float res = fmaf(p, mul, base);
    return (q & 2) ? -res : res;
}


// positional encoding of a 3-vector straight into B-fragment order (arrangement of nerf_mlp.h)
template <int PREC, int L, int KSN>
__device__ __forceinline__ void encode_b(const float (&x)[3], int h, typename Ops<PREC>::B (&out)[KSN]) {
    f32x8 v[KSN];
#pragma unroll
    for (int q = 0; q < KSN * 8; ++q) {
        float y;
#ifdef EVD_ABLATE_PE
        if (q < 3 * L) y = x[q % 3] * (float)(1 << (q / 3));
#else
        if (q < 3 * L) y = sin_or_cos(x[q % 3] * (float)(1 << (q / 3)), h);
#endif
        else if (q == 3 * L) y = h ? x[1] : x[0];
        else if (q == 3 * L + 1) y = h ? 0.f : x[2];
        else y = 0.f;
        v[q >> 3][q & 7] = y;
    }
#pragma unroll
    for (int j = 0; j < KSN; ++j) out[j] = Ops<PREC>::make_b(v[j]);
}

// encode_b with the frequency count L at run time (3 L + 2 <= 8 KSN): the same values at the same positions, the rest zero
template <int PREC, int KSN>
__device__ __forceinline__ void encode_b_rt(const float (&x)[3], int h, int L, typename Ops<PREC>::B (&out)[KSN]) {
    f32x8 v[KSN];
    const int q0 = 3 * L;
#pragma unroll
    for (int q = 0; q < KSN * 8; ++q) {
        float y = 0.f;
        if (q < q0) y = sin_or_cos(x[q % 3] * (float)(1 << (q / 3)), h);
        else if (q == q0) y = h ? x[1] : x[0];
        else if (q == q0 + 1) y = h ? 0.f : x[2];
        v[q >> 3][q & 7] = y;
    }
#pragma unroll
    for (int j = 0; j < KSN; ++j) out[j] = Ops<PREC>::make_b(v[j]);
}

}  // namespace evd
