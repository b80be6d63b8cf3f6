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
// weight stream: chunk c of the packed fragments is resident in LDS slot c&1 while chunk c+1 travels
// global -> registers (issued at the start of chunk c) -> LDS (committed at its end, then one barrier).
template <int PREC> struct Stream {
    static constexpr int NT = mlp_threads(PREC);
    static constexpr int CB = chunk_bytes(PREC);
    static constexpr int FB = frag_bytes(PREC);
    static constexpr int FPC = frags_per_chunk(PREC);
    const char* g;
    char* lds;
    int nchunks, cur, tid;
    f32x4 stg[4];
    __device__ __forceinline__ void issue(int c) {
        if (c < nchunks) {
            const char* src = g + (size_t)c * CB + tid * 16;
#pragma unroll
            for (int t = 0; t < 4; ++t) stg[t] = *reinterpret_cast<const f32x4*>(src + t * NT * 16);
        }
    }
    __device__ __forceinline__ void commit(int c) {
        if (c < nchunks) {
            char* dst = lds + (c & 1) * CB + tid * 16;
#pragma unroll
            for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(dst + t * NT * 16) = stg[t];
        }
    }
    __device__ __forceinline__ void start(const char* gsrc, char* l, int n, int t) {
        g = gsrc; lds = l; nchunks = n; tid = t; cur = 0;
        issue(0);
        commit(0);
        __syncthreads();
    }
    // fragment fc of the resident chunk, this lane's 16 bytes
    __device__ __forceinline__ const char* frag(int fc, int lane) const { return lds + (cur & 1) * CB + fc * FB + lane * 16; }
    __device__ __forceinline__ void chunk_begin() { issue(cur + 1); }
    __device__ __forceinline__ void chunk_end() {
        commit(cur + 1);
        __syncthreads();
        ++cur;
    }
};

enum { OUT_B = 1, OUT_F32 = 2, OUT_BOTH = 3 };

// One linear layer on the wavefront's 32 samples.  in[KSTEPS] are B fragments; the output is either the next
// layer's B fragments (OUT_B: out[2*TILES]) or the raw float32 D fragment of the (single) tile (OUT_F32).
// FOFF = fragment offset inside the current chunk at entry (static); the layer leaves the stream at
// (FOFF + TILES*KSTEPS) % FPC, or chunk-aligned when PAD_END.
template <int PREC, int KSTEPS, int TILES, bool RELU, int OUT, int FOFF, bool PAD_END>
__device__ __forceinline__ void layer(Stream<PREC>& st, const typename Ops<PREC>::B (&in)[KSTEPS],
                                      typename Ops<PREC>::B* __restrict__ out, float* __restrict__ out_f32,
                                      const float* __restrict__ bias, int lane, float* __restrict__ feat_row, int W) {
    typedef Ops<PREC> O;
    constexpr int FPC = Stream<PREC>::FPC;
    constexpr int G = TILES >= 2 ? 2 : 1;
    static_assert(TILES % G == 0, "tile count must be a multiple of the accumulation group");
    const int h = lane >> 5;
#pragma unroll
    for (int p = 0; p < TILES / G; ++p) {
        f32x16 acc[G], accx[G];
#pragma unroll
        for (int t = 0; t < G; ++t) {
            const float* bt = bias + (p * G + t) * 32 + 4 * h;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bt + 8 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) { acc[t][4 * q + e] = bv[e]; accx[t][4 * q + e] = 0.f; }
            }
        }
#pragma unroll
        for (int j = 0; j < KSTEPS; ++j) {
#pragma unroll
            for (int t = 0; t < G; ++t) {
                const int f = FOFF + (p * KSTEPS + j) * G + t;      // static after unrolling
                const int fc = f % FPC;
                if (fc == 0) st.chunk_begin();
                const typename O::A a = O::load_a(st.frag(fc, lane));
                O::mma(acc[t], accx[t], a, in[j]);
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
                out[2 * tile] = O::make_b(v[0]);
                out[2 * tile + 1] = O::make_b(v[1]);
            }
        }
    }
    if (PAD_END && ((FOFF + TILES * KSTEPS) % FPC) != 0) st.chunk_end();
}

// sin(a) for h == 0, cos(a) for h == 1, with one shared code path: Cody-Waite reduction to [-pi/4, pi/4]
// (3 fmaf terms, good for |a| < 1e5) and a quadrant-selected minimax polynomial; ocml beyond that.
__device__ __forceinline__ float sin_or_cos(float a, int h) {
    if (fabsf(a) > 1.0e5f || !(a == a)) return h ? cosf(a) : sinf(a);
    const float j = rintf(a * 0.636619772f);
    float r = fmaf(-j, 1.57079601e+00f, a);
    r = fmaf(-j, 3.13916473e-07f, r);
    r = fmaf(-j, 5.39030253e-15f, r);
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
        if (q < 3 * L) y = sin_or_cos(x[q % 3] * (float)(1 << (q / 3)), h);
        else if (q == 3 * L) y = h ? x[1] : x[0];
        else if (q == 3 * L + 1) y = h ? 0.f : x[2];
        else y = 0.f;
        v[q >> 3][q & 7] = y;
    }
#pragma unroll
    for (int j = 0; j < KSN; ++j) out[j] = Ops<PREC>::make_b(v[j]);
}

}  // namespace evd
