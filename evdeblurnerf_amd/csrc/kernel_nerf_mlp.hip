// Fused NeRF backbone: pts = o + d z -> positional encodings -> 8x256 MLP (skip + view branch) -> raw.
// reference: networks/nerf.py:46-72 (mlpforward), :131-162 (eval), networks/embedding.py:88-98, renderer.py:180.
//
// Design (see nerf_mlp.h for the fragment/permutation contract):
//   * one wavefront owns 32 samples for the whole network; activations never leave registers
//     (the MFMA D fragment of a layer IS the B fragment of the next one),
//   * weights are the A operand: one pre-permuted stream of MFMA fragments per precision, staged
//     global -> registers -> LDS in double-buffered chunks shared by all wavefronts of the workgroup
//     (one barrier per chunk), read back conflict-free with ds_read_b128 (lane-linear fragments),
//   * three arithmetic modes share the code: bf16 (32x32x16), split-float16 x3 (32x32x16, hi/lo operands,
//     2^11-scaled cross terms in a second accumulator) and exact float32 (32x32x2).
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

enum { OUT_B = 0, OUT_F32 = 1 };

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
                if (OUT == OUT_F32) out_f32[r] = x;
            }
            if (feat_row) {          // float32 feature rows: features 32 tile + 8q + 4h + (0..3)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 w4 = {v[q >> 1][(q & 1) * 4], v[q >> 1][(q & 1) * 4 + 1], v[q >> 1][(q & 1) * 4 + 2], v[q >> 1][(q & 1) * 4 + 3]};
                    *reinterpret_cast<f32x4*>(feat_row + 32 * tile + 8 * q + 4 * h) = w4;
                }
            }
            if (OUT == OUT_B) {
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

template <int PREC, int W>
__global__ __launch_bounds__(mlp_threads(PREC), PREC == EVD_PREC_BF16 ? 2 : 1) void k_nerf_mlp(const MlpParams p) {
    typedef Ops<PREC> O;
    typedef typename O::B B;
    constexpr int NT = mlp_threads(PREC);
    constexpr int T = W / 32, KS = W / 16;
    constexpr int FPC = Stream<PREC>::FPC;
    static_assert(W % 64 == 0, "W must be a multiple of 64");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, h = lane >> 5;
    const long s = (long)blockIdx.x * (NT / 2) + wave * 32 + n;
    const bool valid = s < p.nsamp;
    const long sc = valid ? s : p.nsamp - 1;
    const long ray = sc / p.S;
    const float* rb = p.ray_batch + ray * p.ncol;
    const float zv = p.z[sc];
    float pts[3], vd[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        pts[c] = __fadd_rn(rb[c], __fmul_rn(rb[3 + c], zv));   // renderer.py:180
        vd[c] = rb[8 + c];
    }

    // positional encodings straight into B-fragment order (nerf_mlp.h)
    B in_pe[PE_KS], in_dir[PEV_KS];
    {
        f32x8 v[PE_KS];
#pragma unroll
        for (int q = 0; q < PE_KS * 8; ++q) {
            float x;
            if (q < 3 * PE_L) x = sin_or_cos(pts[q % 3] * (float)(1 << (q / 3)), h);
            else if (q == 3 * PE_L) x = h ? pts[1] : pts[0];
            else if (q == 3 * PE_L + 1) x = h ? 0.f : pts[2];
            else x = 0.f;
            v[q >> 3][q & 7] = x;
        }
#pragma unroll
        for (int j = 0; j < PE_KS; ++j) in_pe[j] = O::make_b(v[j]);
        f32x8 w[PEV_KS];
#pragma unroll
        for (int q = 0; q < PEV_KS * 8; ++q) {
            float x;
            if (q < 3 * PE_LV) x = sin_or_cos(vd[q % 3] * (float)(1 << (q / 3)), h);
            else if (q == 3 * PE_LV) x = h ? vd[1] : vd[0];
            else if (q == 3 * PE_LV + 1) x = h ? 0.f : vd[2];
            else x = 0.f;
            w[q >> 3][q & 7] = x;
        }
#pragma unroll
        for (int j = 0; j < PEV_KS; ++j) in_dir[j] = O::make_b(w[j]);
    }

    Stream<PREC> st;
    st.start(p.wstream, smem, p.nchunks, tid);
    const float* bias = p.bias;
    float* frow = (p.feature && valid) ? p.feature + s * W : nullptr;

    B act[KS], nxt[KS];
    layer<PREC, PE_KS, T, true, OUT_B, 0, true>(st, in_pe, nxt, nullptr, bias, lane, (p.feature_kind == 2 && p.D == 1) ? frow : nullptr, W);
    bias += T * 32;
#pragma unroll
    for (int j = 0; j < KS; ++j) act[j] = nxt[j];

#pragma nounroll
    for (int l = 1; l < p.D; ++l) {
        float* fr = (p.feature_kind == 2 && l == p.D - 1) ? frow : nullptr;
        if (l - 1 == p.skip) {      // h = cat([input_pts, h]) feeds this layer (nerf.py:137-138)
            B wide[PE_KS + KS];
#pragma unroll
            for (int j = 0; j < PE_KS; ++j) wide[j] = in_pe[j];
#pragma unroll
            for (int j = 0; j < KS; ++j) wide[PE_KS + j] = act[j];
            layer<PREC, PE_KS + KS, T, true, OUT_B, 0, true>(st, wide, nxt, nullptr, bias, lane, fr, W);
        } else {
            layer<PREC, KS, T, true, OUT_B, 0, true>(st, act, nxt, nullptr, bias, lane, fr, W);
        }
        bias += T * 32;
#pragma unroll
        for (int j = 0; j < KS; ++j) act[j] = nxt[j];
    }

    // heads: alpha_linear (1 tile), feature_linear, views_linears.0 on cat([feature, dirs]), rgb_linear (nerf.py:144-157)
    constexpr int F_ALPHA = KS, F_FEAT = T * KS, F_VIEWS = (T / 2) * (KS + PEV_KS);
    constexpr int OFF_FEAT = F_ALPHA % FPC, OFF_VIEWS = (F_ALPHA + F_FEAT) % FPC, OFF_RGB = (F_ALPHA + F_FEAT + F_VIEWS) % FPC;
    float araw[16], rraw[16];
    layer<PREC, KS, 1, false, OUT_F32, 0, false>(st, act, nullptr, araw, bias, lane, nullptr, W);
    bias += 32;
    layer<PREC, KS, T, false, OUT_B, OFF_FEAT, false>(st, act, nxt, nullptr, bias, lane, p.feature_kind == 1 ? frow : nullptr, W);
    bias += T * 32;
    {
        B vin[KS + PEV_KS];
#pragma unroll
        for (int j = 0; j < KS; ++j) vin[j] = nxt[j];
#pragma unroll
        for (int j = 0; j < PEV_KS; ++j) vin[KS + j] = in_dir[j];
        layer<PREC, KS + PEV_KS, T / 2, true, OUT_B, OFF_VIEWS, false>(st, vin, act, nullptr, bias, lane, nullptr, W);
        bias += (T / 2) * 32;
    }
    {
        B hin[KS / 2];
#pragma unroll
        for (int j = 0; j < KS / 2; ++j) hin[j] = act[j];
        layer<PREC, KS / 2, 1, false, OUT_F32, OFF_RGB, true>(st, hin, nullptr, rraw, bias, lane, nullptr, W);
    }
    if (h == 0 && valid) {
        const f32x4 o = {rraw[0], rraw[1], rraw[2], araw[0]};   // cat([rgb, alpha]) nerf.py:157
        *reinterpret_cast<f32x4*>(p.raw + s * 4) = o;
    }
}

template <int PREC, int W>
static int launch_mlp(const MlpParams& p, hipStream_t st) {
    constexpr int NT = mlp_threads(PREC);
    const long blocks = cdiv(p.nsamp, NT / 2);
    const size_t lds = 2 * (size_t)chunk_bytes(PREC);
    hipLaunchKernelGGL((k_nerf_mlp<PREC, W>), dim3((unsigned)blocks), dim3(NT), lds, st, p);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

int nerf_mlp_dispatch(int prec, int W, const MlpParams& p, hipStream_t st) {
#define EVD_CASE(P, WW) if (prec == P && W == WW) return launch_mlp<P, WW>(p, st)
    EVD_CASE(EVD_PREC_BF16, 256);
    EVD_CASE(EVD_PREC_F16X3, 256);
    EVD_CASE(EVD_PREC_F32, 256);
    EVD_CASE(EVD_PREC_BF16, 64);
    EVD_CASE(EVD_PREC_F16X3, 64);
    EVD_CASE(EVD_PREC_F32, 64);
#undef EVD_CASE
    return fail(EVD_E_INVALID, "evd_nerf_mlp: no kernel for precision %d, width %d (built: W in {64,256})", prec, W);
}

}  // namespace evd
