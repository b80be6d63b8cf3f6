// Software-pipelined fused-MLP machinery of the COMPENSATED float16 mode (EVD_PREC_F16C).
//
// Arithmetic.  Every linear layer is  D = W x  with
//     W x  ~=  f16(W) f16(x)                       v_mfma_f32_32x32x16_f16, as in the float16 mode
//           +  fp6(W - f16(W)) fp6(f16(x))         v_mfma_scale_f32_32x32x64_f8f6f4 (e2m3 operands, e8m0 block scales)
//           +  fp6(W)          fp6(x - f16(x))     the same instruction
// all three accumulating into the same float32 registers.  The float16 product carries the value, the two fp6 products remove the
// rounding error of either operand down to the fp6 resolution of the RESIDUALS (2^-4 of 2^-11): operand error ~2^-15 instead of
// 2^-11, for 1 + 2 x (1/4 the k-steps at 1.25x the rate) = 1.4x the matrix-pipe time of the float16 mode (the split-float16 mode,
// three float16 products, costs 3x).  tools/precision_anatomy.py is the CPU emulation this design was chosen with:
// RGB L-inf vs float64 on trained weights 2.9e-4 (f16) -> 7.6e-6 (this mode); tools/probes/mx_fp6_probe2.hip pins the hardware
// semantics used here (operand k order of the two conversions, scale byte selection, rounding).
//
// Structure (differences to mlp_pipe.h, whose weight ring PStream is reused):
//   * one wavefront per SIMD (256-thread workgroups, up to 512 registers per lane): a layer's input lives in registers three times
//     (float16 fragments, fp6 of the values, fp6 of the residuals);
//   * tile groups of TWO output tiles: the 2 x 16 float32 results a lane holds after a group are exactly one 32-value fp6 block of the
//     next layer (k-steps 4b .. 4b+3).  The group's epilogue (ReLU, float16 convert, residual IN PLACE in the accumulator registers,
//     running maximum for the block scale, then v_cvt_scalef32_pk32_fp6_f16 on the four new fragments and
//     v_cvt_scalef32_2xpk16_fp6_f32 on the two accumulators) is issued between the MFMAs of the next group, also across layers;
//   * the weight stream interleaves, per group and block, 4 x G float16 fragments with the G x 2 fp6 operands (24 bytes per lane each,
//     stored as a 16-byte and an 8-byte part so that both are read with aligned ds_read_b128 / ds_read_b64): 7 G KiB per block;
//     row scales (one e8m0 byte per weight row, layer and fp6 operand) sit next to the biases in LDS.
#pragma once

#include <cstdlib>

#include "mlp_pipe.h"

namespace evd {

// developer ablations (tools/ablate_c.sh compiles variants): 1 / 2 skip the first / second fp6 product, 4 drain every chunk end fully,
// 8 no weight DMA, 16 no barrier, 32 no epilogue, 64 no float16 fragment reads, 128 no fp6 operand reads, 256 no fp6 conversions, 512 no epilogue pairs,
// 1024 the second fp6 operand is not read (the first one is used twice)
#ifdef EVD_C_ABL
constexpr int kAbl = EVD_C_ABL;
#else
constexpr int kAbl = 0;
#endif

// pointers into LDS that keep their address space when they are rebuilt from a (laundered) byte offset: ds_read with immediate offsets
typedef const __attribute__((address_space(3))) float* lds_f32_p;
typedef const __attribute__((address_space(3))) f32x4* lds_f32x4_p;
typedef const __attribute__((address_space(3))) unsigned* lds_u32_p;

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x6 __attribute__((ext_vector_type(6)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));

struct CCfg {
    static constexpr int NT = 256, NW = 4, CB = PIPE_CB, FB = 1024, UPC = CB / 1024, PIECES = CB / 1024 / NW;
#ifdef EVD_C_NSLOT
    static constexpr int NSLOT = EVD_C_NSLOT;
#else
    static constexpr int NSLOT = 4;                          // ring slots: 2 chunks resident, NSLOT - 2 in flight (8 slots measured: no faster, the kernel is LDS-bandwidth-bound)
#endif
#ifdef EVD_C_BARP
    static constexpr int BARP = EVD_C_BARP;
#else
    static constexpr int BARP = 1;                           // chunks per barrier (2 needs NSLOT >= 8)
#endif
    static constexpr int PDM = 4;                            // ring of prefetched float16 A fragments: PDM - 1 MFMAs ahead (8: no faster)
    static constexpr int SAMPLES = NW * 32;
    static constexpr int RING = NSLOT * CB;
    static constexpr int BIAS_WORDS = 5120;                  // biases (32 floats per tile) followed by the row-scale words (32 per tile)
    static constexpr int TOTAL = RING + BIAS_WORDS * 4;
    static_assert((NSLOT & (NSLOT - 1)) == 0 && NSLOT >= 4 && NSLOT - BARP >= BARP + 1 && NSLOT - BARP >= 3 && TOTAL <= 160 * 1024, "ring geometry");
};

// workgroups of a persistent launch of the f16c kernels: one per CU (one wavefront per SIMD; EVD_C_BLOCKS overrides, 0 = one per tile)
static inline int c_persistent_blocks() {
    static const int n = [] {
        if (const char* e = getenv("EVD_C_BLOCKS")) { const int v = atoi(e); return v > 0 ? v : 0x7fffffff; }
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        return cus;
    }();
    return n;
}
static inline long cmin_l(long a, long b) { return a < b ? a : b; }

// Weight stream of this mode: the ring protocol of mlp_pipe.h's PStream (two chunks resident, counted vmcnt, one barrier per 16 KiB
// chunk) with a deeper ring and a cheaper issue.  NSLOT slots: at the top of chunk c (behind the barrier that ended chunk c-1) chunks
// c and c+1 are resident, c+2 .. c+NSLOT-2 in flight, the slot of c-1 is free; chunk_begin(c) issues chunk c+NSLOT-1 into it,
// chunk_end(c) waits until this wavefront's pieces of chunk c+2 have landed (everything younger stays in flight) and crosses the barrier.
// With one wavefront per SIMD nothing else hides the L2 -> LDS latency (under load well above the 2 chunk-times a 4-slot ring gives).
// Issue: the chunk's source is  SGPR base + 32-bit lane offset  (one v_add per chunk instead of a 64-bit add per piece); M0 is set and
// not restored (nothing else in the kernel reads M0): 8 instructions per chunk and wavefront.
template <int NCH> struct CStream {
    static constexpr int kChunks = NCH, NSLOT = CCfg::NSLOT, BARP = CCfg::BARP;
    // BARP chunks per barrier: the barrier behind chunk c (c + 1 a multiple of BARP) releases the slots of every chunk <= c and must find
    // chunks c + 1 .. c + BARP + 1 landed (chunk c + BARP reads ahead into c + BARP + 1 before the next barrier); the DMA runs AHEAD =
    // NSLOT - BARP chunks in front, into the slot of chunk c - BARP, which the last barrier has released whatever c's phase.
    static constexpr int AHEAD = NSLOT - BARP;
    const char* gbase;      // stream base (wave-uniform)
    const char* rd_base;    // ring base + 16 * lane (fragment reads)
    unsigned voff;          // this lane's byte offset of piece 0 of chunk 0
    unsigned dst0;          // LDS byte offset of this wavefront's piece 0 in slot 0 (wave-uniform)
#ifdef EVD_C_STAMP          // developer build: shader-clock cycles this wavefront spends in the vmcnt wait / in the barrier of chunk_end
    long long tw = 0, tb = 0;
#endif
    static_assert(CCfg::PIECES == 4, "four 1 KiB pieces per wavefront per chunk");
    __device__ __forceinline__ void issue(int c) {          // c is a compile-time constant at every call site
        const unsigned off = voff + (unsigned)c * CCfg::CB;
        const unsigned dst = dst0 + (unsigned)(c & (NSLOT - 1)) * CCfg::CB;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                     "global_load_lds_dwordx4 %0, %1 offset:2048\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072"
                     : : "v"(off), "s"(gbase), "s"(dst) : "memory");      // M0 is left changed: hipcc keeps nothing in M0 in this kernel (no other user)
    }
    // one 1 KiB piece k of chunk c (the pieces of a chunk go out one at a time, three consumed units apart: the LDS takes the DMA writes in
    // four short bursts between the fragment reads instead of one long one)
    __device__ __forceinline__ void issue_piece(int c, int k) {
#ifdef EVD_C_M0_PER_PIECE
        const unsigned off = voff + (unsigned)c * CCfg::CB + (unsigned)k * 1024;
        const unsigned dst = dst0 + (unsigned)(c & (NSLOT - 1)) * CCfg::CB + (unsigned)k * 1024;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(gbase), "s"(dst) : "memory");
#else
        // M0 (the LDS base of the chunk's slot) is set by the chunk's first piece and stays: nothing else in these kernels touches M0, and the
        // instruction offset moves the global and the LDS address together (as in issue()) -- 6 scalar instructions per chunk fewer in a
        // kernel whose issue slots are as full as its matrix pipe (measured on one box: 0.904 -> 0.890 ms)
        // ... and M0 for chunk c is written BEHIND the last piece of chunk c - 1 (the pieces go out in stream order; start_issue /
        // restart_issue leave M0 at the first chunk issued here), so that no piece waits on the s_mov -> LDS-DMA hazard slot
        const unsigned off = voff + (unsigned)c * CCfg::CB;
        const unsigned dst_next = dst0 + (unsigned)((c + 1) & (NSLOT - 1)) * CCfg::CB;
        if (k == 0) asm volatile("global_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(gbase) : "memory");
        else if (k == 1) asm volatile("global_load_lds_dwordx4 %0, %1 offset:1024" : : "v"(off), "s"(gbase) : "memory");
        else if (k == 2) asm volatile("global_load_lds_dwordx4 %0, %1 offset:2048" : : "v"(off), "s"(gbase) : "memory");
        else asm volatile("global_load_lds_dwordx4 %0, %1 offset:3072\n\ts_mov_b32 m0, %2" : : "v"(off), "s"(gbase), "s"(dst_next) : "memory");
#endif
    }
    __device__ __forceinline__ void set_m0_for(int c) {       // M0 = LDS base of chunk c's slot (see issue_piece)
        const unsigned dst = dst0 + (unsigned)(c & (NSLOT - 1)) * CCfg::CB;
        asm volatile("s_mov_b32 m0, %0" : : "s"(dst) : "memory");
    }
    // wait until at most `chunks` chunks (PIECES loads each) of this wavefront are outstanding.
    // (Round 4, built and removed: the kernel's own loads in flight across chunk ends -- a persistent kernel fetching its next tile's
    // inputs behind the current tile's last layers.  The VM counter retires in issue order, so every wait for a DMA piece issued BEHIND such
    // a load implies the load: a feature row from the Infinity Cache must land within the ring's AHEAD - 1 chunk times or the stall just
    // moves into the layer (measured: prologue -3.7 k cycles, colour layer 0 +4.0 k), and hipcc drains the queue wherever it copies a
    // destination register of a load it tracks (a deeper ring ran 6x slower for that).)
    static __device__ __forceinline__ void wait_chunks(int chunks) {
        switch (chunks) {
        case 0: wait_vmcnt<0>(); break;
        case 1: wait_vmcnt<4>(); break;
        case 2: wait_vmcnt<8>(); break;
        case 3: wait_vmcnt<12>(); break;
        case 4: wait_vmcnt<16>(); break;
        case 5: wait_vmcnt<20>(); break;
        case 6: wait_vmcnt<24>(); break;
        default: wait_vmcnt<28>(); break;
        }
    }
    __device__ __forceinline__ void start_issue(const char* g, char* ring, int tid) {
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
        gbase = g;
        voff = wave * (CCfg::PIECES * 1024) + lane * 16;
        rd_base = ring + lane * 16;
        dst0 = __builtin_amdgcn_readfirstlane(lds_offset_of(ring) + wave * (CCfg::PIECES * 1024));
#pragma unroll
        for (int c = 0; c < AHEAD; ++c)
            if (c < NCH) issue(c);
        set_m0_for(AHEAD);
    }
    // a persistent workgroup's next pass over the stream; the caller guarantees that every wavefront has crossed the barrier of the last chunk
    __device__ __forceinline__ void restart_issue() {
#pragma unroll
        for (int c = 0; c < AHEAD; ++c)
            if (c < NCH) issue(c);
        set_m0_for(AHEAD);
    }
    __device__ __forceinline__ void start_wait() {      // chunks 0 .. BARP landed
        wait_chunks(cmax(0, cmin(AHEAD, NCH) - (BARP + 1)));
        __syncthreads();
    }
    __device__ __forceinline__ void chunk_begin(int c) { if (c + AHEAD < NCH && !(kAbl & 8)) issue(c + AHEAD); }
    __device__ __forceinline__ void chunk_end(int c) {
        if ((c + 1) % BARP != 0) return;
        if (kAbl & 4) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#ifdef EVD_C_STAMP
        const long long t0 = __builtin_readcyclecounter();
#endif
        wait_chunks((kAbl & 8) ? 0 : cmax(0, cmin(c + AHEAD, NCH - 1) - (c + BARP + 1)));
#ifdef EVD_C_STAMP
        const long long t1 = __builtin_readcyclecounter();
#endif
        if (!(kAbl & 16)) __builtin_amdgcn_s_barrier();
#ifdef EVD_C_STAMP
        const long long t2 = __builtin_readcyclecounter();
        tw += t1 - t0; tb += t2 - t1;
#endif
        asm volatile("" ::: "memory");
    }
};

// one k-block (4 k-steps = 64 input features) of a layer's input as a lane holds it
struct XBlk {
    u32x16 h;        // the four float16 B fragments
    i32x8 qh, ql;    // fp6 (e2m3) of the float16 values / of the residuals x - f16(x); words 6, 7 unused
    unsigned sc;     // byte 0: e8m0 scale of qh, byte 1: of ql
};

__device__ __forceinline__ f16x8 xblk_frag(const XBlk& x, int jj) {
    u32x4 w;
    w[0] = x.h[4 * jj]; w[1] = x.h[4 * jj + 1]; w[2] = x.h[4 * jj + 2]; w[3] = x.h[4 * jj + 3];
    return __builtin_bit_cast(f16x8, w);
}

// running maximum of the float16 magnitudes of a block (as packed 16-bit integers: for non-negative halves integer order = float order)
template <bool SIGNED> __device__ __forceinline__ unsigned c_max_acc(unsigned m, unsigned w) {
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    if (SIGNED) w &= 0x7fff7fffu;
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u16x2, m), __builtin_bit_cast(u16x2, w)));
}

// Block scales from the packed maximum: with E the exponent of the largest float16 magnitude, the values are divided by 2^(E-2)
// (largest in [4, 8): the top of the e2m3 range, 7.5) and the residuals (truncation: |r| < 2^(E-10); EVD_C_RNE: <= 2^(E-11)) by
// 2^(E-12) (2^(E-13)).  The e8m0 byte of a scale is
// the exponent field of the float32 the conversion instructions take.  Returns byte 0 = values, byte 1 = residuals.
__device__ __forceinline__ unsigned c_scales(unsigned m) {
    const unsigned ma = m & 0xffffu, mb = m >> 16;
    const unsigned mm = ma > mb ? ma : mb;                        // max of the two halves
    const unsigned e16 = (mm >> 10) & 31u;                        // biased float16 exponent (0 for zero / subnormal blocks)
    const unsigned bh = e16 + 110u;                               // (e16 - 15) - 2 + 127
#ifdef EVD_C_RNE
    return bh | ((bh - 11u) << 8);
#else
    return bh | ((bh - 10u) << 8);
#endif
}

// finish a block: fp6 of the four float16 fragments and of the 2 x 16 residuals
__device__ __forceinline__ void c_finish(XBlk& x, unsigned m, const f32x16& r0, const f32x16& r1) {
    const unsigned sc = c_scales(m);
    x.sc = sc;
    const float sh = __builtin_bit_cast(float, (sc & 255u) << 23), sl = __builtin_bit_cast(float, ((sc >> 8) & 255u) << 23);
    // inline asm with an EARLY-CLOBBER destination: these multi-pass conversions write their first result registers before they have
    // read their last sources, and hipcc (builtin form) is free to overlap the two (it did: v[18:23] <- v[18:33], v[34:49] -- garbage)
    i32x6 qh, ql;
    asm("v_cvt_scalef32_pk32_fp6_f16 %0, %1, %2" : "=&v"(qh) : "v"(x.h), "v"(sh));
    asm("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=&v"(ql) : "v"(r0), "v"(r1), "v"(sl));
#pragma unroll
    for (int e = 0; e < 6; ++e) { x.qh[e] = qh[e]; x.ql[e] = ql[e]; }
    x.qh[6] = x.qh[7] = x.ql[6] = x.ql[7] = 0;
}

// one epilogue unit: accumulator values 2k, 2k+1 of a tile -> ReLU, float16 pair into the block, residuals in place.
// The float16 pair is TRUNCATED (v_cvt_pkrtz_f16_f32): the residual x - f16(x) then has the sign of x, so for a ReLU layer the
// activation is one v_pk_max_i16 on the pair (a negative half is a negative int16) plus the [0, 1] clamp of the residual's
// v_fma_mix_f32 (x < 0: f16 -> 0, residual = x -> clamped to 0; x >= 0: residual in [0, ulp) is untouched) -- 5 VALU per pair.  The
// truncation error of the float16 product is what the second fp6 product removes; its residuals are one bit larger (c_scales).
// -DEVD_C_RNE: round to nearest + explicit ReLU of the two floats (6 VALU): RGB error 9.5e-6 instead of 1.8e-5 on the trained weights,
// kernel 3-5 % slower.
// The pair is (value v of the group's tile 0, value v of tile 1): the block's float16 elements then run (t0 v0, t1 v0, t0 v1, ...), which is
// the order v_cvt_scalef32_2xpk16_fp6_f32 reads its two sources in -- so both fp6 operands of the block (and the weight operands that pair
// with them, pack.h) index k like the float16 fragments do.
template <bool RELU>
__device__ __forceinline__ void c_drain_pair(f32x16& a0, f32x16& a1, int v, XBlk& x, unsigned& m) {
    float x0 = a0[v], x1 = a1[v];
#ifdef EVD_C_RNE
    if (RELU) { x0 = relu_f32(x0); x1 = relu_f32(x1); }
    const f32x2 xv = {x0, x1};
    const unsigned w = __builtin_bit_cast(unsigned, __builtin_convertvector(xv, f16x2));     // MODE.FP16_OVFL: saturates at +-65504
    constexpr bool CLAMP = false;
#else
    unsigned w = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x0, x1));
    if (RELU) {
        // running maximum on the RAW pair as signed 16-bit integers (a negative half is a negative int16 and never raises m >= 0): it does
        // not read the ReLU's result, so it fills the wait state hipcc puts between a packed instruction and the first reader of its
        // result (544 s_nop per wavefront in a kernel whose issue slots are full)
        const s16x2 zero = {0, 0};
        const unsigned wraw = w;
        w = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, wraw), zero));
        m = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, m), __builtin_bit_cast(s16x2, wraw)));
    }
    constexpr bool CLAMP = RELU;
#endif
    x.h[v] = w;
#ifdef EVD_C_RNE
    m = c_max_acc<!RELU>(m, w);
#else
    if (!RELU) m = c_max_acc<true>(m, w);
#endif
    float r0, r1;                                                  // x - float(f16(x)): v_fma_mix_f32 reads the half straight from the pair
    if (CLAMP) {
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0] clamp" : "=v"(r0) : "v"(w), "v"(x0));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp" : "=v"(r1) : "v"(w), "v"(x1));
    } else {
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(w), "v"(x0));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(w), "v"(x1));
    }
    a0[v] = r0;
    a1[v] = r1;
}

// Static description of one layer.
//   NBLK    k-blocks of the input; LASTK k-steps of the last one (4, or 2: the direction-encoding block of the views layer)
//   TILES   32-row output tiles in groups of G (2; 1 for the float32 heads)        RELU / F32OUT as in mlp_pipe.h
//   CHUNK0  first chunk of the layer (every layer is chunk-aligned)               PAR  accumulator set of the first group
//   PG      tiles of the pending group handed over by the previous layer (0 / 2); PRELU its activation; PDB the input block it becomes
//   NEXT_G  group size of the next layer (0: last layer)                          NCHUNKS chunks of this layer
//   training kernels (TRAIN): OSLOT  first fragment slot of this layer's output in the activation store (-1: not stored); PSLOT slot of
//   the pending block handed over by the previous layer; MSLOT >= 0: the ReLU pattern of this layer's output is collected as a bit mask;
//   PMSLOT >= 0: the mask fragment of the PREVIOUS layer's output is stored there once its pending block is complete.  The store is
//   the one of the single-product float16 mode (mlp_pipe.h act_store, masks as frag_bits): the backward kernels are that mode's.
template <int NBLK_, int LASTK_, int TILES_, int G_, bool RELU_, bool F32OUT_, int CHUNK0_, int PAR_, int PG_, bool PRELU_, int PDB_, int NEXT_G_,
          int OSLOT_ = -1, int PSLOT_ = -1, int MSLOT_ = -1, int PMSLOT_ = -1, int PFRAG0_ = -1>
struct CLayer {
    static constexpr int NBLK = NBLK_, LASTK = LASTK_, TILES = TILES_, G = G_, CHUNK0 = CHUNK0_, PAR = PAR_, PG = PG_, PDB = PDB_, NEXT_G = NEXT_G_;
    static constexpr int OSLOT = OSLOT_, PSLOT = PSLOT_, MSLOT = MSLOT_, PMSLOT = PMSLOT_;
    // index of the pending block's first fragment inside the PRODUCING layer's output (its byte of the mask fragment): 4 x the block the
    // pending group becomes, unless this layer's input blocks are not the producer's output blocks (the skip layer: [h .. | pe | h_last])
    static constexpr int PFRAG0 = PFRAG0_ >= 0 ? PFRAG0_ : 4 * PDB_;
    static constexpr bool RELU = RELU_, F32OUT = F32OUT_, PRELU = PRELU_;
    static constexpr int NG = TILES_ / G_;
    static constexpr int KSTEPS = 4 * (NBLK_ - 1) + LASTK_;
    static constexpr int nk(int b) { return b == NBLK_ - 1 ? LASTK_ : 4; }
    static constexpr int blk_units(int b) { return nk(b) * G_ + 3 * G_; }                  // 1 KiB units of block b of one group
    static constexpr int blk_slots(int b) { return nk(b) * G_ + 2 * G_; }                  // MFMAs
    static constexpr int GROUP_UNITS = (NBLK_ - 1) * (4 * G_ + 3 * G_) + LASTK_ * G_ + 3 * G_;
    static constexpr int GROUP_SLOTS = (NBLK_ - 1) * (4 * G_ + 2 * G_) + LASTK_ * G_ + 2 * G_;
    static constexpr int UNITS = NG * GROUP_UNITS;
    static constexpr int NCHUNKS = cceil(UNITS, CCfg::UPC);
    static constexpr int PAR_OUT = (PAR_ + NG) & 1;
    // unit (from the layer start) of float16 fragment mi of group p = the count of units consumed when its MFMA slot starts
    static constexpr int main_pos(int p, int mi) { return p * GROUP_UNITS + (mi / (4 * G_)) * 7 * G_ + mi % (4 * G_); }
    static constexpr int NMAIN = KSTEPS * TILES_;
    static_assert(TILES_ % G_ == 0 && (G_ == 1 || G_ == 2), "groups of one or two tiles");
    static_assert(NMAIN % CCfg::PDM == 0, "the float16 fragment ring keeps its phase across layers");
    static_assert(UNITS % CCfg::UPC == 0 || UNITS % CCfg::UPC >= 2, "a layer's last chunk must reach its chunk_begin");
};

// register state that flows from layer to layer
struct CPipe {
    f16x8 am[CCfg::PDM];     // ring of prefetched float16 A fragments
    i32x8 ac[2][2];          // fp6 A operands of the current block: [kind][tile of the group]
    f32x16 acc[2][2];        // two accumulator sets of up to two tiles
    unsigned wsc[2][2];      // row-scale words of the tiles in acc[set][t]
    unsigned m;              // running maximum of the block being drained
    unsigned mbits[4];       // training kernels: bit mask of the layer output being completed (CLayer MSLOT)
};

// Training kernels: the activation store of one 32-sample tile as the wavefront addresses it -- a wave-uniform base in SGPRs + this
// lane's 16-byte column + an immediate: fragment slot s lives at base + 1024 s.  (With plain 64-bit lane pointers hipcc computes the ~70
// store addresses of a pass up front and keeps them in VGPRs: the kernel spilled its accumulators to scratch.)  One SGPR base per 4 KiB
// (the instruction's immediate offset reaches 4095), derived by scalar adds.
struct CAct {
    const char* base;       // wave-uniform: store + tile * TILE_BYTES
    unsigned voff;          // lane * 16
};
// (s_nop 1: a VMEM store of more than 8 bytes must be two wait states away from a VALU write of its data registers -- hipcc pads its own
// stores, it cannot see inside the asm; without it the last quad of every 16 lanes stored the NEXT fragment's first word)
__device__ __forceinline__ void c_act_store(const CAct& a, int slot, const u32x4& v) {        // slot is a compile-time constant at every call site
    const char* sb = a.base + (long)(slot >> 2) * 4096;
    switch (slot & 3) {
    case 0: asm volatile("global_store_dwordx4 %0, %1, %2" EVD_ACT_NT_ASM "\n\ts_nop 1" : : "v"(a.voff), "v"(v), "s"(sb) : "memory"); break;
    case 1: asm volatile("global_store_dwordx4 %0, %1, %2 offset:1024" EVD_ACT_NT_ASM "\n\ts_nop 1" : : "v"(a.voff), "v"(v), "s"(sb) : "memory"); break;
    case 2: asm volatile("global_store_dwordx4 %0, %1, %2 offset:2048" EVD_ACT_NT_ASM "\n\ts_nop 1" : : "v"(a.voff), "v"(v), "s"(sb) : "memory"); break;
    default: asm volatile("global_store_dwordx4 %0, %1, %2 offset:3072" EVD_ACT_NT_ASM "\n\ts_nop 1" : : "v"(a.voff), "v"(v), "s"(sb) : "memory"); break;
    }
}

// A finished HIDDEN block (64 channels drained from two output tiles in (tile 0, tile 1) pairs) goes to the activation
// store as the FOUR fragments the single-product float16 mode keeps for the same channels (fragment 2 t + f of tile t = values 8 f .. 8 f + 7
// of the tile's accumulator as pairs, mlp_pipe.h drain_pair): one v_perm_b32 per word picks the tile's halves out of two pairs.  fi0 =
// index of the block's first fragment inside the producing layer's output (the byte of the mask fragment, mlp_pipe.h frag_bits).
template <bool MASK>
__device__ __forceinline__ void c_store_block(const CAct& act, int slot, const XBlk& x, unsigned (&mbits)[4], int fi0) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            unsigned w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = __builtin_amdgcn_perm(x.h[8 * f + 2 * e + 1], x.h[8 * f + 2 * e], tt ? 0x07060302u : 0x05040100u);
            const int fo = 2 * tt + f;
            const u32x4 v = {w[0], w[1], w[2], w[3]};
            c_act_store(act, slot + fo, v);
            if (MASK) mbits[(fi0 + fo) >> 2] |= bits_of_words(w) << (8 * ((fi0 + fo) & 3));
        }
    }
}
// an INPUT block (features / encodings: k-steps in the float16 mode's own order): its four (NK) fragments as they are
template <int NK> __device__ __forceinline__ void c_store_input(const CAct& act, int slot, const XBlk& x) {
#pragma unroll
    for (int jj = 0; jj < NK; ++jj) {
        const u32x4 v = {x.h[4 * jj], x.h[4 * jj + 1], x.h[4 * jj + 2], x.h[4 * jj + 3]};
        c_act_store(act, slot + jj, v);
    }
}

// byte offset of unit u of a layer (units count from the layer's first chunk) inside the ring, for this lane's rd_base
template <class L> __device__ __forceinline__ constexpr int c_ring_off(int u) { return ((L::CHUNK0 + u / CCfg::UPC) & (CCfg::NSLOT - 1)) * CCfg::CB + (u % CCfg::UPC) * 1024; }

// One tile group P of layer L.  in[]: input blocks; the pending group of the previous layer is drained into in[L::PDB]; this layer's
// groups are drained into out[p].  bias: LDS bias block of this layer (tile-major, 32 floats per tile) with the row-scale words
// SC_OFF words behind it.  NXT: the next layer (for the prefetch across the layer boundary), void for the last.
template <class L, class NXT, class ST, int NIN, int NOUT, int P, bool TRAIN = false>
__device__ __forceinline__ void c_group(ST& st, CPipe& pp, XBlk (&in)[NIN], XBlk (&out)[NOUT], lds_f32_p bias, int lane, const CAct& act = CAct{}) {
    constexpr int G = L::G, NS = L::GROUP_SLOTS;
    constexpr int cur = (L::PAR + P) & 1, oth = cur ^ 1;
    constexpr bool FIRST = P == 0, LAST = P == L::NG - 1;
    constexpr int DG = FIRST ? L::PG : G;                       // tiles to drain
    constexpr int QU = DG > 0 ? DG * 8 + 1 : 0;                 // pair units + the finishing unit
    constexpr int NBL0 = (P == L::NG - 1 ? L::NEXT_G : G) * 5;
    constexpr int s_dep = (FIRST && L::PG > 0) ? L::PDB * (4 * G + 2 * G) : NS;     // first slot that needs the drained block
    constexpr int dend = cmin(s_dep, NS - cceil(NBL0, 3) - 1);       // the epilogue is spread over the whole group; the bias reads follow it
    constexpr bool drain_first = QU > 0 && dend - 2 < 1;
    constexpr int drate = (QU > 0 && !drain_first) ? cceil(QU, dend - 2) : (drain_first ? QU : 0);
    constexpr int BG = LAST ? L::NEXT_G : G;                    // tiles whose bias / scales are fetched for the next group
    constexpr int NBL = BG * 5;                                 // 4 bias reads + 1 scale word per tile
    constexpr int b0 = cmin(drain_first ? 0 : dend - 1, NS - 1);
    constexpr int brate = NBL > 0 ? cceil(NBL, NS - b0) : 0;
    auto units_thru = [](int s) constexpr { return (QU == 0 || s < 0) ? 0 : (drain_first ? QU : (s < 2 ? 0 : cmin(QU, ((s - 1) * QU) / (dend - 2)))); };
    auto bias_thru = [](int s) constexpr { return (NBL == 0 || s <= b0) ? 0 : (s >= NS ? NBL : cmin(NBL, (s - b0) * brate)); };
    const int h = lane >> 5;
    lds_f32_p bias_next = bias + (LAST ? L::TILES : (P + 1) * G) * 32;
    constexpr int SC_OFF = CCfg::BIAS_WORDS / 2;
    constexpr int ubase = P * L::GROUP_UNITS;                   // first unit of this group
    constexpr int mbase = P * L::KSTEPS * G;                    // first float16 MFMA of this group (ring phase)

    // one filler step: everything that is issued in front of MFMA slot s
    auto filler = [&](int s) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < drate; ++i) {
            const int u = units_thru(s - 1) + i;
            if (u < units_thru(s) && !(kAbl & 32)) {
                XBlk& dst = FIRST ? in[L::PDB] : out[P > 0 ? P - 1 : 0];
                if (u < DG * 8) {
                    if (kAbl & 512) continue;
                    static_assert(DG == 0 || DG == 2, "blocks are drained from groups of two tiles");
                    if (u == 0) pp.m = 0u;
                    if (FIRST) c_drain_pair<L::PRELU>(pp.acc[oth][0], pp.acc[oth][1], u, dst, pp.m);
                    else c_drain_pair<L::RELU>(pp.acc[oth][0], pp.acc[oth][1], u, dst, pp.m);
                } else if (!(kAbl & 256)) {
                    c_finish(dst, pp.m, pp.acc[oth][0], pp.acc[oth][1]);
                    if constexpr (TRAIN) {
                        if constexpr (FIRST) {
                            if constexpr (L::PSLOT >= 0) {
                                c_store_block<(L::PMSLOT >= 0)>(act, L::PSLOT, dst, pp.mbits, L::PFRAG0);
                                if constexpr (L::PMSLOT >= 0) {      // the producing layer's output is complete: its mask fragment
                                    const u32x4 mv = {pp.mbits[0], pp.mbits[1], pp.mbits[2], pp.mbits[3]};
                                    c_act_store(act, L::PMSLOT, mv);
                                    pp.mbits[0] = pp.mbits[1] = pp.mbits[2] = pp.mbits[3] = 0u;
                                }
                            }
                        } else if constexpr (L::OSLOT >= 0) {
                            c_store_block<(L::MSLOT >= 0)>(act, L::OSLOT + 4 * (P > 0 ? P - 1 : 0), dst, pp.mbits, 4 * (P > 0 ? P - 1 : 0));
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < brate; ++i) {
            const int b = bias_thru(s - 1) + i;
            if (b < bias_thru(s)) {
                const int bt = b / 5, q = b % 5;
                if (q < 4) {
                    const f32x4 bv = *(lds_f32x4_p)(bias_next + bt * 32 + 8 * q + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) pp.acc[oth][bt][4 * q + e] = bv[e];
                } else {
                    pp.wsc[oth][bt] = ((lds_u32_p)bias_next)[SC_OFF + bt * 32 + (lane & 31)];
                }
            }
        }
    };
    // chunk protocol after a slot that raised the count of consumed units from t0 to t1: the DMA of chunk c + 3 follows the second unit
    // of chunk c (the matrix pipe has restarted behind the barrier by then), the barrier that releases chunk c its last unit
    auto chunks = [&](int t0, int t1) __attribute__((always_inline)) {
        const int c0 = t0 / CCfg::UPC, c1 = t1 / CCfg::UPC;
#ifndef EVD_C_NOSPREAD      // (all four pieces at the chunk's second unit: 1.4 % slower)
        auto due = [](int t) constexpr { return cmin(CCfg::PIECES, (t + 1) / 3); };      // pieces issued once t units of the chunk are consumed: at 2, 5, 8, 11
        auto pieces = [&](int c, int k0, int k1) __attribute__((always_inline)) {
            if (c + ST::AHEAD >= ST::kChunks || (kAbl & 8)) return;
#pragma unroll
            for (int k = 0; k < CCfg::PIECES; ++k)
                if (k >= k0 && k < k1) st.issue_piece(c + ST::AHEAD, k);
        };
        if (c0 == c1) {
            pieces(L::CHUNK0 + c0, due(t0 % CCfg::UPC), due(t1 % CCfg::UPC));
        } else {
            pieces(L::CHUNK0 + c0, due(t0 % CCfg::UPC), CCfg::PIECES);
            st.chunk_end(L::CHUNK0 + c0);
            pieces(L::CHUNK0 + c1, 0, due(t1 % CCfg::UPC));
        }
#else
        if (c0 == c1) {
            if (t0 % CCfg::UPC < 2 && t1 % CCfg::UPC >= 2) st.chunk_begin(L::CHUNK0 + c0);
        } else {
            st.chunk_end(L::CHUNK0 + c0);
            if (t1 % CCfg::UPC >= 2) st.chunk_begin(L::CHUNK0 + c1);
        }
#endif
    };

    int s = 0, ub = ubase, mm = mbase;      // all three are compile-time constants after unrolling
#pragma unroll
    for (int b = 0; b < L::NBLK; ++b) {
        const int nk = L::nk(b);
        const int lo16 = ub + nk * G, hi8 = lo16 + 2 * G;        // unit offsets of the fp6 parts of this block
        const int ncl = 4 * G;                                  // fp6 part loads of this block: (kind, tile) x (16-byte, 8-byte part)
        const int per_slot = cceil(ncl, nk * G);
#pragma unroll
        for (int jj = 0; jj < nk; ++jj) {
#pragma unroll
            for (int t = 0; t < G; ++t) {
                const int i = jj * G + t;                       // main slot inside the block
                filler(s);
                {   // float16 fragment of the main MFMA PDM - 1 ahead (possibly in the next block, group or layer)
                    constexpr int LA = CCfg::PDM - 1;
                    const int idx = mm + LA - mbase;            // main index inside this group
                    int u = -1, ringoff = 0;
                    if (idx < L::KSTEPS * G) {
                        u = L::main_pos(P, idx);
                        ringoff = c_ring_off<L>(u);
                    } else if (!LAST) {
                        u = L::main_pos(P + 1, idx - L::KSTEPS * G);
                        ringoff = c_ring_off<L>(u);
                    } else if constexpr (!std::is_void<NXT>::value) {
                        // the next layer starts in the chunk behind this layer's last one, which is resident only once that last
                        // chunk is the current one; fragments wanted earlier are fetched by c_layer after the layer's last barrier
                        if ((ub + i) / CCfg::UPC >= L::NCHUNKS - 1) {
                            u = NXT::main_pos(0, idx - L::KSTEPS * G);
                            ringoff = c_ring_off<NXT>(u);
                        }
                    }
                    if (u >= 0 && !(kAbl & 64)) pp.am[(mm + LA) % CCfg::PDM] = *reinterpret_cast<const f16x8*>(st.rd_base + ringoff);
                }
#pragma unroll
                for (int q = 0; q < per_slot; ++q) {            // fp6 parts of this block, consumed by its last 2 G slots
                    const int c = i * per_slot + q;
                    if (c < ncl && !(kAbl & 128) && !((kAbl & 1024) && (c >> 1) / G == 1)) {      // (1024: the second fp6 operand is not read: what deriving it from the float16 fragments would save)
                        const int kt = c >> 1, part = c & 1;    // kt = kind * G + tile
                        const int kind = kt / G, tt = kt % G;
                        if (part == 0) {
                            const u32x4 w = *reinterpret_cast<const u32x4*>(st.rd_base + c_ring_off<L>(lo16 + kt));
#pragma unroll
                            for (int e = 0; e < 4; ++e) pp.ac[kind][tt][e] = (int)w[e];
                        } else {
                            const int uo = hi8 + kt / 2;
                            const u32x2 w = *reinterpret_cast<const u32x2*>(st.rd_base + c_ring_off<L>(uo) - 8 * lane + (kt & 1) * 512);
                            pp.ac[kind][tt][4] = (int)w[0];
                            pp.ac[kind][tt][5] = (int)w[1];
                            pp.ac[kind][tt][6] = 0;
                            pp.ac[kind][tt][7] = 0;
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                pp.acc[cur][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pp.am[mm % CCfg::PDM], xblk_frag(in[b], jj), pp.acc[cur][t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                chunks(ub + i, ub + i + 1);
                ++s;
                ++mm;
            }
        }
#pragma unroll
        for (int kind = 0; kind < 2; ++kind) {
#pragma unroll
            for (int t = 0; t < G; ++t) {
                const int i = kind * G + t;
                filler(s);
                __builtin_amdgcn_sched_barrier(0);
                // kind 0: fp6(Wl) x fp6(f16(x)), scale bytes 0 / 0; kind 1: fp6(W) x fp6(x - f16(x)), scale bytes 1 / 1
                // (the two scale operands pinned to VGPRs: with the larger register footprint of the TRAIN variant hipcc's AGPR rewrite
                // otherwise parks a block's scale word in an AGPR and then emits the MFMA with it -- "Operand has incorrect register class")
                int sca = (int)pp.wsc[cur][t], scb = (int)in[b].sc;
                if (TRAIN) asm("" : "+v"(sca), "+v"(scb));
                if (kind == 0 && (kAbl & 1)) {}
                else if (kind == 1 && (kAbl & 2)) {}
                else if (kind == 0)
                    pp.acc[cur][t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(pp.ac[0][t], in[b].qh, pp.acc[cur][t], 2, 2, 0, sca, 0, scb);
                else
                    pp.acc[cur][t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(pp.ac[(kAbl & 1024) ? 0 : 1][t], in[b].ql, pp.acc[cur][t], 2, 2, 1, sca, 1, scb);
                __builtin_amdgcn_sched_barrier(0);
                const int last = 2 * G - 1;
                chunks(lo16 + i, i == last ? hi8 + G : lo16 + i + 1);
                ++s;
            }
        }
        ub = hi8 + G;
    }
#pragma unroll
    for (int i = 0; i < NBL; ++i) {                             // bias rows that did not fit between the MFMAs
        const int bq = bias_thru(NS - 1) + i;
        if (bq < NBL) {
            const int bt = bq / 5, q = bq % 5;
            if (q < 4) {
                const f32x4 bv = *(lds_f32x4_p)(bias_next + bt * 32 + 8 * q + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) pp.acc[oth][bt][4 * q + e] = bv[e];
            } else {
                pp.wsc[oth][bt] = ((lds_u32_p)bias_next)[SC_OFF + bt * 32 + (lane & 31)];
            }
        }
    }
}

template <class L, class NXT, class ST, int NIN, int NOUT, int P, bool TRAIN = false> struct CGroupLoop {
    static __device__ __forceinline__ void run(ST& st, CPipe& pp, XBlk (&in)[NIN], XBlk (&out)[NOUT], lds_f32_p bias, int lane, const CAct& act = CAct{}) {
        c_group<L, NXT, ST, NIN, NOUT, P, TRAIN>(st, pp, in, out, bias, lane, act);
        if constexpr (P + 1 < L::NG) CGroupLoop<L, NXT, ST, NIN, NOUT, P + 1, TRAIN>::run(st, pp, in, out, bias, lane, act);
    }
};

// One linear layer on the wavefront's 32 samples.  `out` receives the blocks of every group but the last, which stays pending in the
// accumulators (F32OUT: rows 0..3 of the single tile are returned in out_f32).
template <class L, class NXT, class ST, int NIN, int NOUT, bool TRAIN = false>
__device__ __forceinline__ void c_layer(ST& st, CPipe& pp, XBlk (&in)[NIN], XBlk (&out)[NOUT], float* out_f32, lds_f32_p bias, int lane, const CAct& act = CAct{}) {
    static_assert(NIN >= L::NBLK, "input blocks");
    CGroupLoop<L, NXT, ST, NIN, NOUT, 0, TRAIN>::run(st, pp, in, out, bias, lane, act);
    if (L::UNITS % CCfg::UPC != 0) {             // the zero-padded tail of the layer's last chunk
#ifndef EVD_C_NOSPREAD
        constexpr int c = L::CHUNK0 + L::UNITS / CCfg::UPC, k0 = cmin(CCfg::PIECES, (L::UNITS % CCfg::UPC + 1) / 3);
        if (c + ST::AHEAD < ST::kChunks && !(kAbl & 8)) {
#pragma unroll
            for (int k = k0; k < CCfg::PIECES; ++k) st.issue_piece(c + ST::AHEAD, k);
        }
#endif
        st.chunk_end(L::CHUNK0 + L::UNITS / CCfg::UPC);
    }
    if constexpr (!std::is_void<NXT>::value) {       // first fragments of the next layer that the last group could not prefetch (c_group)
#pragma unroll
        for (int k = 0; k < CCfg::PDM - 1; ++k)
            if (L::main_pos(L::NG - 1, L::KSTEPS * L::G - (CCfg::PDM - 1) + k) / CCfg::UPC < L::NCHUNKS - 1)
                pp.am[k] = *reinterpret_cast<const f16x8*>(st.rd_base + c_ring_off<NXT>(NXT::main_pos(0, k)));
    }
    if (L::F32OUT) {
        constexpr int cur = (L::PAR + L::NG - 1) & 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            out_f32[r] = pp.acc[cur][0][r];
            asm volatile("" : "+v"(out_f32[r]));       // pin (mlp_pipe.h): else hipcc sinks the head's MFMA chain to the end of the kernel
        }
    }
}

// prologue: first PDM - 1 float16 fragments of the first layer, bias and row scales of its first group
template <class L, class ST> __device__ __forceinline__ void c_prime(ST& st, CPipe& pp, lds_f32_p bias, int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int i = 0; i < CCfg::PDM - 1; ++i) pp.am[i] = *reinterpret_cast<const f16x8*>(st.rd_base + c_ring_off<L>(L::main_pos(0, i)));
#pragma unroll
    for (int t = 0; t < L::G; ++t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bv = *(lds_f32x4_p)(bias + t * 32 + 8 * q + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) pp.acc[L::PAR][t][4 * q + e] = bv[e];
        }
        pp.wsc[L::PAR][t] = ((lds_u32_p)bias)[CCfg::BIAS_WORDS / 2 + t * 32 + (lane & 31)];
    }
    pp.m = 0u;
    pp.mbits[0] = pp.mbits[1] = pp.mbits[2] = pp.mbits[3] = 0u;
}

// sin(2 pi t) / cos(2 pi t) of the revolution count t = (frac + lo): v_sin_f32 takes revolutions
__device__ __forceinline__ float c_sin_rev(float thi, float tlo, float scale, int h) {
    // 2^k t: the product with the high part and its fract are exact; the low part rides along
    const float f = __builtin_amdgcn_fractf(thi * scale);
    return __builtin_amdgcn_sinf(fmaf(tlo, scale, f + (h ? 0.25f : 0.f)));
}

// positional encoding of a 3-vector as one input block (arrangement of nerf_mlp.h: position q = 8 j + e of lane half h), all three
// representations.  KSN = 4 (point encoding) or 2 (direction encoding: the upper half of the block is zero).
template <int L, int KSN> __device__ __forceinline__ void c_encode(const float (&x)[3], int h, XBlk& out) {
    float thi[3], tlo[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {      // x / 2 pi in two floats
        thi[c] = x[c] * 0.15915494309189535f;                                                  // the float nearest 1 / 2 pi ...
        tlo[c] = fmaf(x[c], 0.15915494309189535f, -thi[c]) + x[c] * 6.4206383e-09f;            // ... is 6.42e-9 below it
    }
    f32x16 r[2];
    unsigned m = 0u;
#pragma unroll
    for (int q = 0; q < 32; q += 2) {
        float y[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int qq = q + i;
            if (qq >= KSN * 8) y[i] = 0.f;
            else if (qq < 3 * L) y[i] = c_sin_rev(thi[qq % 3], tlo[qq % 3], (float)(1 << (qq / 3)), h);
            else if (qq == 3 * L) y[i] = h ? x[1] : x[0];
            else if (qq == 3 * L + 1) y[i] = h ? 0.f : x[2];
            else y[i] = 0.f;
        }
        r[0][q >> 1] = y[0];           // element i = position q of the encoding: pair v = q / 2 is (r[0][v], r[1][v]), as in the layers' epilogue
        r[1][q >> 1] = y[1];
    }
#pragma unroll
    for (int v = 0; v < 16; ++v) c_drain_pair<false>(r[0], r[1], v, out, m);
    c_finish(out, m, r[0], r[1]);
}

}  // namespace evd
