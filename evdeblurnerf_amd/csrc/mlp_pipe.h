// Software-pipelined fused-MLP machinery (third generation) for networks whose layer table is known at compile time.
// (Generic-depth networks, the float32 / split-float16 modes and the PDRF kernels use mlp_device.h's `layer`.)
//
// What changed against mlp_device.h, and why (numbers: profiles/README.md):
//   * The epilogue of a tile group (bias is already in the accumulator; ReLU, convert to the next layer's B
//     fragments) no longer runs between the groups with the matrix pipe idle.  Accumulators are double-buffered
//     and the previous group's epilogue is cut into 2-value units that are issued BETWEEN the MFMAs of the
//     current group -- also across layer boundaries: a layer's last group is drained inside the first group of
//     the next layer, whose k-steps are ordered so that it needs those inputs last.
//   * The bias of the next group is read from LDS straight into the idle accumulator set during the second half
//     of the current group; the first MFMA of a group therefore accumulates onto the bias and no VALU is spent.
//   * The weight ring has 4 slots of 16 KiB and the protocol keeps TWO chunks resident, so A fragments are
//     prefetched across chunk (and layer) boundaries and nothing waits for LDS after a barrier.
//   * A wavefront may own NS = 2 sample tiles (64 samples): every A fragment read from LDS then feeds two MFMAs.
#pragma once

#include <type_traits>

#include "mlp_device.h"

namespace evd {

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

constexpr int PIPE_CB = 16384;      // chunk bytes of the pipelined streams (all precisions)
constexpr int pipe_fpc(int prec) { return PIPE_CB / frag_bytes(prec); }

// max(x, 0) as ONE v_max_f32 (fmaxf() costs two: hipcc first canonicalises the operand, a signalling-NaN nicety;
// here NaN in -> 0 out, where torch's relu propagates the NaN -- activations are finite)
__device__ __forceinline__ float relu_f32(float x) {
#ifdef EVD_PIPE_RELU_C
    return fmaxf(x, 0.f);
#else
    float y;
    asm("v_max_f32_e32 %0, 0, %1" : "=v"(y) : "v"(x));
    return y;
#endif
}

// ---------------------------------------------------------------------------------------------
// precision policies with pair-wise B-fragment construction (set_pair<RELU>: activation, range guard, convert)
template <int PREC> struct POps;

struct MaskFrag { unsigned w[4]; };      // one lane's 16 bytes of a bit-mask fragment (frag_bits / mask_from_bits below)
__device__ __forceinline__ unsigned mask_word(unsigned g, unsigned a);
__device__ __forceinline__ unsigned bits_of_words(const unsigned (&w)[4]);
__device__ __forceinline__ unsigned mask_from_bits(const MaskFrag& m, int fo, int e);

// Every policy also says how the TRAINING kernels treat a B fragment: mask_act (gradient . [saved activation != 0]), mask_bits (the
// same from a bit-mask fragment), bits (the ReLU pattern of a completed fragment as one byte), zero.
template <> struct POps<EVD_PREC_BF16> {
    struct B { unsigned w[4]; };
    static __device__ __forceinline__ void mask_act(B& g, const B& a) {
#pragma unroll
        for (int e = 0; e < 4; ++e) g.w[e] = mask_word(g.w[e], a.w[e]);
    }
    static __device__ __forceinline__ void mask_bits(B& g, const MaskFrag& m, int fo) {
#pragma unroll
        for (int e = 0; e < 4; ++e) g.w[e] &= mask_from_bits(m, fo, e);
    }
    static __device__ __forceinline__ unsigned bits(const B& f) { return bits_of_words(f.w); }
    static __device__ __forceinline__ void zero(B& b) { b.w[0] = b.w[1] = b.w[2] = b.w[3] = 0u; }
    typedef bf16x8 A;
    static constexpr bool kSplit = false, kFastTrig = true;
    static __device__ __forceinline__ A load_a(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }
    static __device__ __forceinline__ void mma(f32x16& acc, f32x16&, const A& a, const B& b, bool) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
    // RELU on the packed pair: as int16, a negative bfloat16 (sign bit set) is a negative integer, so ONE
    // v_pk_max_i16 against 0 clears both halves' negatives (-0 -> +0); positives are unchanged
    template <bool RELU> static __device__ __forceinline__ void set_pair(B& b, int e, float x0, float x1) {
        const f32x2 v = {x0, x1};
        unsigned w = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
        if (RELU) {
            const s16x2 zero = {0, 0};
            w = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, w), zero));
        }
        b.w[e] = w;
    }
};

template <> struct POps<EVD_PREC_F16> {
    struct B { unsigned w[4]; };
    static __device__ __forceinline__ void mask_act(B& g, const B& a) {
#pragma unroll
        for (int e = 0; e < 4; ++e) g.w[e] = mask_word(g.w[e], a.w[e]);
    }
    static __device__ __forceinline__ void mask_bits(B& g, const MaskFrag& m, int fo) {
#pragma unroll
        for (int e = 0; e < 4; ++e) g.w[e] &= mask_from_bits(m, fo, e);
    }
    static __device__ __forceinline__ unsigned bits(const B& f) { return bits_of_words(f.w); }
    static __device__ __forceinline__ void zero(B& b) { b.w[0] = b.w[1] = b.w[2] = b.w[3] = 0u; }
    typedef f16x8 A;
    static constexpr bool kSplit = false, kFastTrig = true;
    static __device__ __forceinline__ A load_a(const char* p) { return *reinterpret_cast<const f16x8*>(p); }
    static __device__ __forceinline__ void mma(f32x16& acc, f32x16&, const A& a, const B& b, bool) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
    }
    // The kernel sets MODE.FP16_OVFL (pipe_fp16_saturate(), tools/probes/fp16_ovfl_probe.hip): a conversion that overflows
    // saturates to +-65504 instead of inf, so the float16 range guard costs nothing; ReLU is ONE v_pk_max_i16 on the
    // packed pair (a negative float16 is a negative int16), as in the bf16 mode.
    template <bool RELU> static __device__ __forceinline__ void set_pair(B& b, int e, float x0, float x1) {
        const f32x2 v = {x0, x1};
        unsigned w = __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
        if (RELU) {
            const s16x2 zero = {0, 0};
            w = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, w), zero));
        }
        b.w[e] = w;
    }
};

template <> struct POps<EVD_PREC_F16X3> {
    struct B { unsigned hi[4], lo[4]; };
    // a value is hi + lo / 2048: it is zero iff both halves are (a ReLU output has hi >= 0; lo, the scaled residual, has either sign)
    static __device__ __forceinline__ void mask_act(B& g, const B& a) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned any = a.hi[e] | a.lo[e];
            g.hi[e] = mask_word(g.hi[e], any);
            g.lo[e] = mask_word(g.lo[e], any);
        }
    }
    static __device__ __forceinline__ void mask_bits(B& g, const MaskFrag& m, int fo) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned k = mask_from_bits(m, fo, e);
            g.hi[e] &= k;
            g.lo[e] &= k;
        }
    }
    static __device__ __forceinline__ unsigned bits(const B& f) {
        const unsigned any[4] = {f.hi[0] | f.lo[0], f.hi[1] | f.lo[1], f.hi[2] | f.lo[2], f.hi[3] | f.lo[3]};
        return bits_of_words(any);
    }
    static __device__ __forceinline__ void zero(B& b) {
#pragma unroll
        for (int e = 0; e < 4; ++e) b.hi[e] = b.lo[e] = 0u;
    }
    typedef Ops<EVD_PREC_F16X3>::A A;
    static constexpr bool kSplit = true, kFastTrig = false;
    static __device__ __forceinline__ A load_a(const char* p) { return Ops<EVD_PREC_F16X3>::load_a(p); }
    static __device__ __forceinline__ void mma(f32x16& acc, f32x16& accx, const A& a, const B& b, bool first) {
        const f16x8 bh = __builtin_bit_cast(f16x8, b.hi), bl = __builtin_bit_cast(f16x8, b.lo);
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi, bh, acc, 0, 0, 0);
        accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi, bl, first ? zero : accx, 0, 0, 0);
        accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.lo, bh, accx, 0, 0, 0);
    }
    template <bool RELU> static __device__ __forceinline__ void set_pair(B& b, int e, float x0, float x1) {
        const f32x2 c = {__builtin_amdgcn_fmed3f(x0, RELU ? 0.f : -65000.f, 65000.f), __builtin_amdgcn_fmed3f(x1, RELU ? 0.f : -65000.f, 65000.f)};   // ReLU + float16 range guard
        const f16x2 hi = __builtin_convertvector(c, f16x2);
        const f32x2 back = __builtin_convertvector(hi, f32x2);
        const f16x2 lo = __builtin_convertvector((c - back) * 2048.f, f16x2);
        b.hi[e] = __builtin_bit_cast(unsigned, hi);
        b.lo[e] = __builtin_bit_cast(unsigned, lo);
    }
};

template <> struct POps<EVD_PREC_F32> {
    struct B { float v[8]; };
    typedef f32x8 A;
    static constexpr bool kSplit = false, kFastTrig = false;
    static __device__ __forceinline__ A load_a(const char* p) { return Ops<EVD_PREC_F32>::load_a(p); }
    static __device__ __forceinline__ void mma(f32x16& acc, f32x16&, const A& a, const B& b, bool) {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b.v[e], acc, 0, 0, 0);
    }
    template <bool RELU> static __device__ __forceinline__ void set_pair(B& b, int e, float x0, float x1) {
        b.v[2 * e] = RELU ? relu_f32(x0) : x0;
        b.v[2 * e + 1] = RELU ? relu_f32(x1) : x1;
    }
};

// ---------------------------------------------------------------------------------------------
// CB_: chunk bytes of the LDS ring (the packed streams are padded to PIPE_CB = 16 KiB; every layer of the built networks is a
// whole number of 8 KiB chunks too, so a kernel may walk the same stream in 8 KiB chunks: half the ring, two workgroups per CU)
// HI_ONLY_ (training forward of the split-float16 arithmetic in front of the SINGLE-product float16 backward, EVD_PREC_F16C training):
// the activation store is the float16 mode's -- 1 KiB slots holding the hi halves (the value rounded to float16) of the fragments.
template <int PREC, int NS_, int NT_, int CB_ = PIPE_CB, bool HI_ONLY_ = false> struct PipeCfg {
    typedef POps<PREC> O;
    static constexpr bool HI_ONLY = HI_ONLY_;
    static constexpr int STORE_PREC = HI_ONLY_ ? 3 /*EVD_PREC_F16*/ : PREC;
    static_assert(!HI_ONLY_ || PREC == 1 /*EVD_PREC_F16X3*/, "hi-only stores belong to the split-float16 forward");
    static constexpr int PRECISION = PREC, NS = NS_, NT = NT_, NW = NT_ / 64;
    static constexpr int FB = frag_bytes(PREC);
    static constexpr int CB = CB_;
    static constexpr int FPC = CB / FB;
    static constexpr int NSLOT = 4;
    static constexpr int PIECES = CB / 1024 / NW;        // 1 KiB DMA pieces per wavefront per chunk
#ifdef EVD_PIPE_PD
    static constexpr int PD = EVD_PIPE_PD;
#else
    static constexpr int PD = PREC == EVD_PREC_F32 ? 2 : 4;   // A-fragment prefetch depth
#endif
    static constexpr int SAMPLES = NW * NS_ * 32;        // samples per workgroup
    static constexpr int RING = NSLOT * CB;
    static constexpr int BIAS_FLOATS = 4096;
    static constexpr int STASH_FRAGS = PE_KS + PEV_KS;     // per sample tile: point encoding + direction encoding
    static constexpr int STASH_PER_WAVE = NS_ * STASH_FRAGS * 64 * (int)sizeof(typename O::B);
    static constexpr int TOTAL = RING + BIAS_FLOATS * 4 + NW * STASH_PER_WAVE;
    static_assert(PIECES * NW * 1024 == CB && PD <= FPC, "ring geometry");
};

// MODE.FP16_OVFL = 1 for this wavefront (hwreg HW_REG_MODE = 1, bit 23, width 1): float16 results that overflow are
// clamped to +-MAX_FP16 instead of becoming inf.  Called first thing by the kernels of the float16 mode.
template <int PREC> __device__ __forceinline__ void pipe_fp16_saturate() {
    if (PREC == EVD_PREC_F16) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Weight stream, 4-slot ring; chunk c lives in slot c % 4.  Every chunk index is a compile-time constant where it is
// used (the layer table is static), so slots, LDS offsets and the counted waits are immediates and the main loop has
// no branches.  Invariant at the top of chunk c (after the barrier that ended chunk c-1): chunks c and c+1 are
// resident and visible to every wavefront, chunk c+2 is in flight, the slot of chunk c-1 is free.
//   chunk_begin(c): issue the DMA of chunk c+3 into the slot of chunk c-1
//   chunk_end(c):   counted vmcnt -- this wavefront's pieces of chunk c+2 have landed, those of c+3 stay in
//                   flight -- then ONE s_barrier.  LDS reads already issued for chunk c+1 stay in flight.
// STORES: kernels that also issue global stores in the main loop wait for vmcnt(0) (loads and stores share the
// counter and may retire out of order with respect to each other).
template <class C, bool STORES, int NCH> struct PStream {
    const char* gsrc;       // this lane's source address of piece 0 of chunk 0
    const char* rd_base;    // ring base + 16 * lane (fragment reads)
    unsigned dst0;          // LDS byte offset of this wavefront's piece 0 in slot 0 (wave-uniform)
    // The instruction's immediate offset advances BOTH the global source and the LDS destination
    // (tools/probes/glds_offset_probe.hip), so the pieces of a chunk share one M0 set-up and one address.
    __device__ __forceinline__ void issue(int c) {
        const char* src = gsrc + (size_t)c * C::CB;
        const unsigned dst = __builtin_amdgcn_readfirstlane(dst0 + (c & 3) * C::CB);
        unsigned keep;
        static_assert(C::PIECES == 2 || C::PIECES == 4, "1 KiB pieces per wavefront per chunk");
        if (C::PIECES == 2)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        else
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                         "global_load_lds_dwordx4 %1, off offset:2048\n\tglobal_load_lds_dwordx4 %1, off offset:3072\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    }
    static constexpr int kChunks = NCH;
    // one 1 KiB piece k of chunk c (kernels that spread a chunk's DMA over several MFMA gaps, mlp_pipe_c.h)
    __device__ __forceinline__ void issue_piece(int c, int k) {
        const char* src = gsrc + (size_t)c * C::CB + k * 1024;
        const unsigned dst = __builtin_amdgcn_readfirstlane(dst0 + (c & 3) * C::CB + k * 1024);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    }
    // start_issue() goes first in the kernel: the DMA of the first three chunks flies while the prologue stages the
    // biases, loads the rays and computes the positional encodings; start_wait() closes the prologue.
    __device__ __forceinline__ void start_issue(const char* g, char* ring, int tid) {
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
        gsrc = g + wave * (C::PIECES * 1024) + lane * 16;
        rd_base = ring + lane * 16;
        dst0 = lds_offset_of(ring) + wave * (C::PIECES * 1024);
        issue(0);
        if (NCH > 1) issue(1);
        if (NCH > 2) issue(2);
    }
    __device__ __forceinline__ void start_wait() {
        if (NCH > 2 && !STORES) wait_vmcnt<C::PIECES>();
        else wait_vmcnt<0>();
        __syncthreads();
    }
    // fragment fc of chunk c, this lane's 16 bytes
    __device__ __forceinline__ const char* frag(int c, int fc) const { return rd_base + (c & 3) * C::CB + fc * C::FB; }
#ifdef EVD_ABL_DMA
    __device__ __forceinline__ void chunk_begin(int c) {}
#else
    __device__ __forceinline__ void chunk_begin(int c) { if (c + 3 < NCH) issue(c + 3); }
#endif
    __device__ __forceinline__ void chunk_end(int c) {
#ifndef EVD_ABL_DMA
        if (c + 3 < NCH && !STORES) wait_vmcnt<C::PIECES>();
        else wait_vmcnt<0>();
#endif
#ifndef EVD_ABL_BARRIER
        __builtin_amdgcn_s_barrier();
#endif
        asm volatile("" ::: "memory");
    }
};

// The same interface on a stream that is RESIDENT in LDS (round 6: the 64-wide PDRF level -- its whole packed stream is 5 chunks in the
// split-float16 mode, 3 in float16): the persistent workgroup copies all NCH chunks once, after that a chunk boundary costs nothing (no DMA,
// no counted wait, no barrier: the wavefronts of the workgroup run their tiles independently).
template <class C, int NCH> struct PResident {
    const char* rd_base;    // stream base in LDS + 16 * lane
    static constexpr int kChunks = NCH;
    static constexpr int BYTES = NCH * C::CB;
    __device__ __forceinline__ void start_issue(const char* g, char* lds, int tid) {
        rd_base = lds + (tid & 63) * 16;
        for (int i = tid; i < BYTES / 16; i += C::NT) *reinterpret_cast<f32x4*>(lds + (size_t)i * 16) = *reinterpret_cast<const f32x4*>(g + (size_t)i * 16);
    }
    __device__ __forceinline__ void start_wait() { __syncthreads(); }
    __device__ __forceinline__ const char* frag(int c, int fc) const { return rd_base + c * C::CB + fc * C::FB; }
    __device__ __forceinline__ void chunk_begin(int) {}
    __device__ __forceinline__ void chunk_end(int) {}
};

// register state that flows from layer to layer: prefetched A fragments and the two accumulator sets
template <class C> struct Pipe {
    typename C::O::A abuf[C::PD];
    f32x16 acc[2][2][C::NS];
    f32x16 accx[2][2][C::NS];      // cross-term accumulators (split precision only; dead otherwise)
    unsigned mbits[C::NS][4];      // training kernels: bit mask of the layer output being completed (LayerDesc MSLOT)
};

constexpr int cmin(int a, int b) { return a < b ? a : b; }
constexpr int cmax(int a, int b) { return a > b ? a : b; }
constexpr int cceil(int a, int b) { return (a + b - 1) / b; }

// static description of one layer of a fused network
//   KTOT     k-steps (16 input features each)            TILES   32-row output tiles, processed in groups of G
//   RELU     activation of this layer                    F32OUT  the single tile is wanted as float32 (heads)
//   CHUNK0   chunk that FOFF counts from                 FOFF    fragment offset of the layer from the start of CHUNK0
//   PAD_END  the stream is chunk-aligned after this layer
//   AOFF     prefetch-ring phase at entry                PAR     accumulator set of the first group
//   PG       tiles of the pending group handed over by the previous layer (0 = none); PRELU its activation;
//            PDOFF the k-step of `in` its outputs become; PFEAT_TILE0 >= 0: also store it as feature rows
//   OWN_FEAT store this layer's outputs as float32 feature rows
//   NEXT_G   tiles of the next layer's first group (0 = last layer of the kernel: no prefetch beyond)
//   OSLOT    training kernels: fragment slot of this layer's output in the activation store (-1: not stored);
//   PSLOT    ... slot of the FIRST fragment of the pending group handed over by the previous layer
//   MSLOT    ... >= 0: the ReLU pattern of this layer's output is also collected as a bit mask (frag_bits) in pp.mbits; the mask
//            fragment is stored by the NEXT layer when it completes the pending group: its PMSLOT (slot) and PFRAG0 (fragment index
//            of the first pending fragment inside the producing layer)
template <int KTOT_, int TILES_, int G_, bool RELU_, bool F32OUT_, int CHUNK0_, int FOFF_, bool PAD_END_, int AOFF_, int PAR_, int PG_,
          bool PRELU_, int PDOFF_, int PFEAT_TILE0_, bool OWN_FEAT_, int NEXT_G_, int OSLOT_ = -1, int PSLOT_ = -1, int MSLOT_ = -1, int PMSLOT_ = -1,
          int PFRAG0_ = 0>
struct LayerDesc {
    static constexpr int KTOT = KTOT_, TILES = TILES_, CHUNK0 = CHUNK0_, FOFF = FOFF_, AOFF = AOFF_, PAR = PAR_, PG = PG_, PDOFF = PDOFF_,
                         PFEAT_TILE0 = PFEAT_TILE0_, NEXT_G = NEXT_G_, OSLOT = OSLOT_, PSLOT = PSLOT_, MSLOT = MSLOT_, PMSLOT = PMSLOT_,
                         PFRAG0 = PFRAG0_;
    static constexpr bool RELU = RELU_, F32OUT = F32OUT_, PAD_END = PAD_END_, PRELU = PRELU_, OWN_FEAT = OWN_FEAT_;
    static constexpr int G = G_;
    static constexpr int NG = TILES_ / G;
    static constexpr int NF = TILES_ * KTOT_;
    static constexpr int PAR_OUT = (PAR_ + NG) & 1;
    static constexpr int AOFF_OUT = AOFF_ + NF;      // reduce modulo PD at the use
    static_assert(G_ >= 1 && G_ <= 2 && TILES_ % G_ == 0, "tile count must be a multiple of the group size (1 or 2)");
    static_assert(!F32OUT_ || TILES_ == 1, "float32 heads have one tile");
};

// ReLU + convert of two accumulator values into a B-fragment element pair; optionally keeps the float32 values
template <class C, bool RELU>
__device__ __forceinline__ void drain_pair(const f32x16& a, const f32x16& ax, int k, typename C::O::B& dst, float* __restrict__ frow4) {
    float x0 = C::O::kSplit ? fmaf(ax[2 * k], 4.8828125e-4f, a[2 * k]) : a[2 * k];
    float x1 = C::O::kSplit ? fmaf(ax[2 * k + 1], 4.8828125e-4f, a[2 * k + 1]) : a[2 * k + 1];
    C::O::template set_pair<RELU>(dst, k & 3, x0, x1);
    if (frow4) {                      // float32 feature rows: features 32 tile + 8q + 4h + (0..3), q = k / 2
        const f32x2 v = {RELU ? relu_f32(x0) : x0, RELU ? relu_f32(x1) : x1};
        *reinterpret_cast<f32x2*>(frow4 + 2 * (k & 1)) = v;
    }
}

// Static filler schedule of one tile group.  Slot m = the code placed right before MFMA m of the group.
//   drain: the 2-value epilogue units of the previous group (FIRST: the previous layer's pending group).  Nothing in
//          slots 0 and 1 (that group's last MFMAs are still in the pipe); all units issued through slot dend - 1,
//          i.e. before the first MFMA that consumes them (m_dep) and within the first half of the group.
//   bias:  the next group's bias rows, read into the accumulator set that the drain has just freed.
template <class C, class L, bool FIRST, bool LAST> struct GroupSched {
    static constexpr int NS = C::NS, G = L::G, NM = G * NS * L::KTOT;
    static constexpr int DG = FIRST ? L::PG : G;
    static constexpr int QU = DG * NS * 8;
    static constexpr int m_dep = (FIRST && L::PG > 0) ? L::PDOFF * G * NS : NM;
    static constexpr int dend = cmin(m_dep, NM / 2 + 2);
    static constexpr bool drain_first = QU > 0 && dend - 2 < 1;      // no room to overlap: drain before the first MFMA
    static constexpr int drate = (QU > 0 && !drain_first) ? cceil(QU, dend - 2) : (drain_first ? QU : 0);
    static constexpr int BG = LAST ? L::NEXT_G : G;
    static constexpr int NB = BG * 4 * NS;
    static constexpr int b0 = cmin(cmax(drain_first ? 0 : dend - 1, NM / 2), NM - 1);
    static constexpr int brate = NB > 0 ? cceil(NB, NM - b0) : 0;
    // cumulative units / loads issued through slot m (m = -1: nothing yet; m = NM: the tail after the last MFMA)
    static constexpr int units_thru(int m) { return (QU == 0 || m < 0) ? 0 : (drain_first ? QU : (m < 2 ? 0 : cmin(QU, (m - 1) * drate))); }
    static constexpr int bias_thru(int m) { return (NB == 0 || m <= b0) ? 0 : (m >= NM ? NB : cmin(NB, (m - b0) * brate)); }
};

// One tile group p of a layer (see pipe_layer).
// a completed B fragment of the training kernels goes to the activation store (lane-linear: 16 bytes per lane, 1 KiB per fragment;
// the split-float16 mode's fragments are two such halves, hi then lo, in a 2 KiB slot).  FB: bytes of a fragment slot of the store.
// The fragment stores of the training kernels are NON-TEMPORAL (round 6): the store is written once and read again by the backward a
// millisecond later, from HBM either way (2.36 M fine-level samples x 66 KiB per 32 = 4.9 GB per iteration against 32 MB of L2); without the
// cache allocation the fine level's training forward runs 1.446 -> 1.259 ms per iteration, the coarse one 0.231 -> 0.190, the backward
// kernels that read the fragments unchanged (profiles/r06_act_nt_ab.log).  -DEVD_ACT_NO_NT restores the plain stores.
#ifndef EVD_ACT_NO_NT
#define EVD_ACT_ST(ptr, val) __builtin_nontemporal_store((val), (ptr))
#define EVD_ACT_NT_ASM " nt"
#else
#define EVD_ACT_ST(ptr, val) (*(ptr) = (val))
#define EVD_ACT_NT_ASM ""
#endif
template <int FB = 1024, class F> __device__ __forceinline__ void act_store(char* act_lane, int slot, const F& frag) {
    static_assert(sizeof(F) == 16 || (sizeof(F) == 32 && FB == 2048), "16-byte fragments, or hi / lo pairs in 2 KiB slots");
    if constexpr (sizeof(F) == 16) {
        EVD_ACT_ST(reinterpret_cast<f32x4*>(act_lane + (long)slot * FB), __builtin_bit_cast(f32x4, frag));
    } else {
        struct Two { f32x4 a, b; };
        const Two t = __builtin_bit_cast(Two, frag);
        EVD_ACT_ST(reinterpret_cast<f32x4*>(act_lane + (long)slot * FB), t.a);
        EVD_ACT_ST(reinterpret_cast<f32x4*>(act_lane + (long)slot * FB + 1024), t.b);
    }
}

// ... of a kernel with configuration C: its own fragment format, or (C::HI_ONLY) the hi halves in the float16 mode's 1 KiB slots
template <class C, class F> __device__ __forceinline__ void pipe_act_store(char* act_lane, int slot, const F& frag) {
    if constexpr (C::HI_ONLY && sizeof(F) == 32) {
        struct Two { f32x4 a, b; };
        EVD_ACT_ST(reinterpret_cast<f32x4*>(act_lane + (long)slot * 1024), __builtin_bit_cast(Two, frag).a);
    } else if constexpr (C::HI_ONLY) {
        act_store<1024>(act_lane, slot, frag);
    } else {
        act_store<C::FB>(act_lane, slot, frag);
    }
}

// gradient word . [activation != 0], both halves (activations are post-ReLU: masked <=> the stored half is +0)
__device__ __forceinline__ unsigned mask_word(unsigned g, unsigned a) {
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    const u16x2 av = __builtin_bit_cast(u16x2, a), zero = {0, 0};
    return g & __builtin_bit_cast(unsigned, av != zero);
}

// bit e = (element e of the packed fragment != 0): the ReLU pattern of a stored activation fragment in one byte
__device__ __forceinline__ unsigned bits_of_words(const unsigned (&w)[4]) {
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    const u16x2 one = {1, 1};
    unsigned b = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {       // v_pk_min_u16 against 1: (half != 0) in bits 0 and 16
        const unsigned t = __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(u16x2, w[e]), one));
        b |= ((t | (t >> 15)) & 3u) << (2 * e);
    }
    return b;
}
template <class B> __device__ __forceinline__ unsigned frag_bits(const B& f) { return bits_of_words(f.w); }
// packed-pair mask word e of fragment fo from a mask fragment (byte j of a lane's 16 bytes = frag_bits of fragment j)
__device__ __forceinline__ unsigned mask_from_bits(const MaskFrag& m, int fo, int e) {
    const unsigned two = (m.w[fo >> 2] >> (8 * (fo & 3) + 2 * e)) & 3u;
    return ((0u - (two & 1u)) & 0xffffu) | ((0u - (two >> 1)) & 0xffff0000u);
}
template <class B> __device__ __forceinline__ unsigned mask_from_bits(const B& m, int fo, int e) {
    return mask_from_bits(__builtin_bit_cast(MaskFrag, m), fo, e);
}

// OMASK (backward kernels): every stored output fragment is first multiplied by the 0/1 ReLU pattern of the saved activation:
// 1 = from the activation fragments omask[sample tile][fragment], 2 = from one bit-mask fragment omask[sample tile]
template <class C, class L, class ST, int NOUT, int P, bool TRAIN, int OMASK>
__device__ __forceinline__ void pipe_group(ST& st, Pipe<C>& pp, typename C::O::B (&in)[C::NS][L::KTOT],
                                           typename C::O::B (&out)[C::NS][NOUT], const float* __restrict__ bias, int h,
                                           float* const* frow, char* const* act, const void* omask) {
    typedef typename C::O O;
    typedef GroupSched<C, L, P == 0, P == L::NG - 1> S;
    constexpr int FB = C::FB;
    constexpr int NS = C::NS, FPC = C::FPC, PD = C::PD, G = L::G, KTOT = L::KTOT, NF = L::NF, NM = S::NM;
    constexpr int ENDV = L::PAD_END ? cceil(L::FOFF + NF, FPC) * FPC : L::FOFF + NF;
    constexpr int cur = (L::PAR + P) & 1, oth = cur ^ 1;
    constexpr bool FIRST = P == 0;
    const float* bias_next = bias + (P == L::NG - 1 ? L::TILES : (P + 1) * G) * 32 + 4 * h;

#pragma unroll
    for (int j = 0; j < KTOT; ++j) {
#pragma unroll
        for (int t = 0; t < G; ++t) {
            const int f = (P * KTOT + j) * G + t;        // fragment index inside the layer
            const int fc = (L::FOFF + f) % FPC, c = L::CHUNK0 + (L::FOFF + f) / FPC;
#ifdef EVD_PIPE_DMA_EARLY
            if (fc == 0) st.chunk_begin(c);
#endif
            {   // prefetch fragment f + PD - 1 (possibly the next layer's) into the register set freed by step f - 1
                const int idx = f + PD - 1;
#ifdef EVD_ABL_LDS
                if ((idx < NF || L::NEXT_G > 0) && idx < PD) {
#else
                if (idx < NF || L::NEXT_G > 0) {
#endif
                    const int v = idx < NF ? L::FOFF + idx : ENDV + (idx - NF);
                    pp.abuf[(L::AOFF + idx) % PD] = O::load_a(st.frag(L::CHUNK0 + v / FPC, v % FPC));
                }
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int m = (j * G + t) * NS + s;
#pragma unroll
                for (int i = 0; i < S::drate; ++i) {         // epilogue units of the previous group
                    const int u = S::units_thru(m - 1) + i;
#ifdef EVD_ABL_DRAIN
                    if (u < S::units_thru(m) && u < 1) {
#else
                    if (u < S::units_thru(m)) {
#endif
                        const int dt = u / (NS * 8), ds = (u / 8) % NS, k = u % 8;
                        if (FIRST) {
                            float* fr = (L::PFEAT_TILE0 >= 0 && frow[ds]) ? frow[ds] + 32 * (L::PFEAT_TILE0 + dt) + 8 * (k >> 1) + 4 * h : nullptr;
                            drain_pair<C, L::PRELU>(pp.acc[oth][dt][ds], pp.accx[oth][dt][ds], k, in[ds][(L::PG > 0 ? L::PDOFF : 0) + 2 * dt + (k >> 2)], fr);
                            if constexpr (TRAIN && L::PSLOT >= 0)
                                if ((k & 3) == 3) {
                                    const auto& pf = in[ds][(L::PG > 0 ? L::PDOFF : 0) + 2 * dt + (k >> 2)];
                                    pipe_act_store<C>(act[ds], L::PSLOT + 2 * dt + (k >> 2), pf);
                                    if constexpr (L::PMSLOT >= 0) {
                                        const int fi = L::PFRAG0 + 2 * dt + (k >> 2);
                                        pp.mbits[ds][fi >> 2] |= O::bits(pf) << (8 * (fi & 3));
                                        if (dt == L::PG - 1 && k == 7) {         // the producing layer's output is complete: its mask fragment
                                            MaskFrag mf;
#pragma unroll
                                            for (int e = 0; e < 4; ++e) { mf.w[e] = pp.mbits[ds][e]; pp.mbits[ds][e] = 0u; }
                                            pipe_act_store<C>(act[ds], L::PMSLOT, mf);
                                        }
                                    }
                                }
                        } else {
                            constexpr int tile0 = P > 0 ? (P - 1) * G : 0;
                            float* fr = (L::OWN_FEAT && frow[ds]) ? frow[ds] + 32 * (tile0 + dt) + 8 * (k >> 1) + 4 * h : nullptr;
                            drain_pair<C, L::RELU>(pp.acc[oth][dt][ds], pp.accx[oth][dt][ds], k, out[ds][2 * (tile0 + dt) + (k >> 2)], fr);
                            if constexpr (TRAIN && L::OSLOT >= 0)
                                if ((k & 3) == 3) {
                                    const int fo = 2 * (tile0 + dt) + (k >> 2);
                                    if constexpr (OMASK == 1) O::mask_act(out[ds][fo], static_cast<const typename O::B*>(omask)[ds * NOUT + fo]);
                                    else if constexpr (OMASK == 2) O::mask_bits(out[ds][fo], static_cast<const MaskFrag*>(omask)[ds], fo);
                                    pipe_act_store<C>(act[ds], L::OSLOT + fo, out[ds][fo]);
                                    if constexpr (L::MSLOT >= 0) pp.mbits[ds][fo >> 2] |= O::bits(out[ds][fo]) << (8 * (fo & 3));
                                }
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < S::brate; ++i) {         // bias rows of the next group
                    const int b = S::bias_thru(m - 1) + i;
                    if (b < S::bias_thru(m)) {
                        const int bt = b / (4 * NS), bs = (b / 4) % NS, q = b % 4;
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_next + bt * 32 + 8 * q);
#pragma unroll
                        for (int e = 0; e < 4; ++e) pp.acc[oth][bt][bs][4 * q + e] = bv[e];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                O::mma(pp.acc[cur][t][s], pp.accx[cur][t][s], pp.abuf[(L::AOFF + f) % PD], in[s][j], j == 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#ifndef EVD_PIPE_DMA_EARLY
            if (fc == (FPC > 2 ? 1 : 0)) st.chunk_begin(c);      // after the first MFMAs of the chunk: the matrix pipe restarts right behind the barrier
#endif
            if (fc == FPC - 1) st.chunk_end(c);
        }
    }
#pragma unroll
    for (int i = 0; i < S::NB; ++i) {                        // bias rows that did not fit between the MFMAs
        const int b = S::bias_thru(NM - 1) + i;
        if (b < S::NB) {
            const int bt = b / (4 * NS), bs = (b / 4) % NS, q = b % 4;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_next + bt * 32 + 8 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) pp.acc[oth][bt][bs][4 * q + e] = bv[e];
        }
    }
}

template <class C, class L, class ST, int NOUT, int P, bool TRAIN, int OMASK>
struct GroupLoop {
    static __device__ __forceinline__ void run(ST& st, Pipe<C>& pp, typename C::O::B (&in)[C::NS][L::KTOT],
                                               typename C::O::B (&out)[C::NS][NOUT], const float* __restrict__ bias, int h,
                                               float* const* frow, char* const* act, const void* omask) {
        pipe_group<C, L, ST, NOUT, P, TRAIN, OMASK>(st, pp, in, out, bias, h, frow, act, omask);
        if constexpr (P + 1 < L::NG) GroupLoop<C, L, ST, NOUT, P + 1, TRAIN, OMASK>::run(st, pp, in, out, bias, h, frow, act, omask);
    }
};

// One linear layer on the wavefront's NS x 32 samples (see the file header).  `in` holds the B fragments of all
// k-steps except the pending ones, which this layer produces itself while it runs; `out` receives the B fragments
// of every group but the last, which stays pending in the accumulators (or, F32OUT, is returned in out_f32).
// frow[s]: float32 feature row of sample tile s (null lanes = invalid samples), W floats per sample.
template <class C, class L, class ST, int NOUT, bool TRAIN = false, int OMASK = 0>
__device__ __forceinline__ void pipe_layer(ST& st, Pipe<C>& pp, typename C::O::B (&in)[C::NS][L::KTOT],
                                           typename C::O::B (&out)[C::NS][NOUT], float (*out_f32)[4],
                                           const float* __restrict__ bias, int lane, float* const* frow, char* const* act = nullptr,
                                           const void* omask = nullptr) {
    typedef typename C::O O;
    GroupLoop<C, L, ST, NOUT, 0, TRAIN, OMASK>::run(st, pp, in, out, bias, lane >> 5, frow, act, omask);
    if (L::PAD_END && ((L::FOFF + L::NF) % C::FPC) != 0) st.chunk_end(L::CHUNK0 + (L::FOFF + L::NF) / C::FPC);
    if (L::F32OUT) {
        constexpr int cur = (L::PAR + L::NG - 1) & 1;
#pragma unroll
        for (int s = 0; s < C::NS; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                out_f32[s][r] = O::kSplit ? fmaf(pp.accx[cur][0][s][r], 4.8828125e-4f, pp.acc[cur][0][s][r]) : pp.acc[cur][0][s][r];
                // pin: without a side-effecting use HERE hipcc sinks this head's whole MFMA chain to the end of the kernel
                // (where the value is stored) and spills the A fragments it had prefetched in place
                asm volatile("" : "+v"(out_f32[s][r]));
            }
    }
}

// Drain the pending last group of layer L into `out` without overlap (kernels that end with a fragment-producing layer)
template <class C, class L, int NOUT>
__device__ __forceinline__ void pipe_flush(Pipe<C>& pp, typename C::O::B (&out)[C::NS][NOUT]) {
    constexpr int cur = (L::PAR + L::NG - 1) & 1, tile0 = (L::NG - 1) * L::G;
#pragma unroll
    for (int t = 0; t < L::G; ++t)
#pragma unroll
        for (int s = 0; s < C::NS; ++s)
#pragma unroll
            for (int k = 0; k < 8; ++k) drain_pair<C, L::RELU>(pp.acc[cur][t][s], pp.accx[cur][t][s], k, out[s][2 * (tile0 + t) + (k >> 2)], nullptr);
}

// prologue of the pipeline: first PD - 1 fragments of chunk 0 and the bias of the first group of the first layer
template <class C, class L, class ST>
__device__ __forceinline__ void pipe_prime(ST& st, Pipe<C>& pp, const float* __restrict__ bias, int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int s = 0; s < C::NS; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) pp.mbits[s][e] = 0u;
#pragma unroll
    for (int i = 0; i < C::PD - 1; ++i) pp.abuf[(L::AOFF + i) % C::PD] = C::O::load_a(st.frag(L::CHUNK0, L::FOFF + i));
#pragma unroll
    for (int t = 0; t < L::G; ++t)
#pragma unroll
        for (int s = 0; s < C::NS; ++s)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + t * 32 + 8 * q + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) pp.acc[L::PAR][t][s][4 * q + e] = bv[e];
            }
}

// sin (h == 0) / cos (h == 1) on the hardware unit: v_sin_f32 takes revolutions, so cos(a) = sin(a / 2pi + 1/4).
// Absolute error ~ |a| 2^-24 + 2^-20 (|a| < 800 rad here): far below the 2^-8 / 2^-11 operand rounding of the
// bf16 / f16 modes that use it; the float32-grade modes keep sin_or_cos().
__device__ __forceinline__ float sin_or_cos_hw(float a, int h) {
#ifdef EVD_PIPE_EXACT_TRIG
    return sin_or_cos(a, h);
#else
    return __builtin_amdgcn_sinf(fmaf(a, 0.15915494309189535f, h ? 0.25f : 0.f));
#endif
}

// ... with the compensated float16 mode's sines (mlp_pipe_c.h c_sin_rev): x / 2 pi in two floats, 2^k (hi part) exact, its fract exact, the
// hardware unit on the reduced revolution count -- ~1e-6 absolute instead of the float32-grade polynomial's ~1e-7, a fifth of its instructions
template <class C, int L, int KSN>
__device__ __forceinline__ void encode_pairs_rev(const float (&x)[3], int h, typename C::O::B (&out)[KSN]) {
    float thi[3], tlo[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        thi[c] = x[c] * 0.15915494309189535f;
        tlo[c] = fmaf(x[c], 0.15915494309189535f, -thi[c]) + x[c] * 6.4206383e-09f;
    }
#pragma unroll
    for (int q = 0; q < KSN * 8; q += 2) {
        float y[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int qq = q + i;
            if (qq < 3 * L) {
                const float sc = (float)(1 << (qq / 3));
                y[i] = __builtin_amdgcn_sinf(fmaf(tlo[qq % 3], sc, __builtin_amdgcn_fractf(thi[qq % 3] * sc) + (h ? 0.25f : 0.f)));
            } else if (qq == 3 * L) y[i] = h ? x[1] : x[0];
            else if (qq == 3 * L + 1) y[i] = h ? 0.f : x[2];
            else y[i] = 0.f;
        }
        C::O::template set_pair<false>(out[q >> 3], (q & 7) >> 1, y[0], y[1]);
    }
}

// positional encoding of a 3-vector straight into B-fragment order (arrangement of nerf_mlp.h)
template <class C, int L, int KSN>
__device__ __forceinline__ void encode_pairs(const float (&x)[3], int h, typename C::O::B (&out)[KSN]) {
#pragma unroll
    for (int q = 0; q < KSN * 8; q += 2) {
        float y[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int qq = q + i;
            if (qq < 3 * L) y[i] = C::O::kFastTrig ? sin_or_cos_hw(x[qq % 3] * (float)(1 << (qq / 3)), h)
                                                   : sin_or_cos(x[qq % 3] * (float)(1 << (qq / 3)), h);
            else if (qq == 3 * L) y[i] = h ? x[1] : x[0];
            else if (qq == 3 * L + 1) y[i] = h ? 0.f : x[2];
            else y[i] = 0.f;
        }
        C::O::template set_pair<false>(out[q >> 3], (q & 7) >> 1, y[0], y[1]);
    }
}

}  // namespace evd
