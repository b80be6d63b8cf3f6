// Host-side packing of nn.Linear weights into MFMA fragment streams (contract: nerf_mlp.h).
#pragma once

#include <cstdint>
#include <cmath>
#include <cstring>
#include <vector>

#include "evd_common.h"
#include "nerf_mlp.h"

namespace evd {

__host__ __device__ static inline uint16_t f32_to_bf16(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN
    u += 0x7fffu + ((u >> 16) & 1u);                                            // round to nearest even
    return (uint16_t)(u >> 16);
}

// element (lane l, position e) of one fragment at `dst`, every precision's layout (nerf_mlp.h); shared by the host packer and
// the device re-packer (k_pack_stream, evd_api.hip) so that both produce identical bytes
__host__ __device__ static inline void put_element(int prec, uint8_t* dst, int l, int e, float w) {
    if (prec == EVD_PREC_BF16) {
        *reinterpret_cast<uint16_t*>(dst + l * 16 + e * 2) = f32_to_bf16(w);
    } else if (prec == EVD_PREC_F16) {
        *reinterpret_cast<_Float16*>(dst + l * 16 + e * 2) = (_Float16)w;
    } else if (prec == EVD_PREC_F16X3) {
        const _Float16 hi = (_Float16)w;
        *reinterpret_cast<_Float16*>(dst + l * 16 + e * 2) = hi;
        *reinterpret_cast<_Float16*>(dst + 1024 + l * 16 + e * 2) = (_Float16)((w - (float)hi) * 2048.f);
    } else {
        *reinterpret_cast<float*>(dst + (e < 4 ? 0 : 1024) + l * 16 + (e & 3) * 4) = w;
    }
}

struct StreamBuilder {
    int prec;
    int group = 2;                  // tiles per group (kernel's G); 1-tile layers are always their own group
    size_t cb;                      // chunk bytes: layers flagged pad_end are zero-padded to a multiple of it
    std::vector<uint8_t> bytes;
    const float* arena = nullptr;   // when set: every element's source is also recorded as an index into this array
    std::vector<int32_t> src;       // [fragment][lane][position] -> arena index, -1 = zero
    explicit StreamBuilder(int p, size_t chunk = 0) : prec(p), cb(chunk ? chunk : (size_t)chunk_bytes(p)) {}
    // one fragment of output tile `tile`, k-step j.  row(tile, r) -> source row or -1 (zero row); col(j, kk) -> source
    // column or -1 (zero padding); at(row, col) -> address of that weight or null
    template <class RowFn, class ColFn, class AtFn>
    void frag_at(int tile, int j, RowFn row, ColFn col, AtFn at) {
        const int fb = frag_bytes(prec);
        const size_t base = bytes.size();
        bytes.resize(base + fb, 0);
        uint8_t* dst = bytes.data() + base;
        for (int l = 0; l < 64; ++l) {
            const int r = row(tile, l & 31);
            for (int e = 0; e < 8; ++e) {
                const int c = col(j, 8 * (l >> 5) + e);
                const float* w = (r >= 0 && c >= 0) ? at(r, c) : nullptr;
                put_element(prec, dst, l, e, w ? *w : 0.f);
                if (arena) src.push_back(w ? (int32_t)(w - arena) : -1);
            }
        }
    }
    // fragments in kernel order: tile groups of 2 (or 1), k-steps inside, tiles of the group innermost
    template <class RowFn, class ColFn, class AtFn>
    void layer_at(int tiles, int ksteps, bool pad_end, RowFn row, ColFn col, AtFn at) {
        const int G = (tiles % group == 0) ? group : 1;
        for (int p = 0; p < tiles / G; ++p)
            for (int j = 0; j < ksteps; ++j)
                for (int t = 0; t < G; ++t) frag_at(p * G + t, j, row, col, at);
        if (pad_end) pad();
    }
    // row-major weight matrix Wm [*, in_dim]
    template <class RowFn, class ColFn>
    void layer_rc(const float* Wm, int in_dim, int tiles, int ksteps, bool pad_end, RowFn row, ColFn col) {
        layer_at(tiles, ksteps, pad_end, row, col, [=](int r, int c) { return c < in_dim ? Wm + (size_t)r * in_dim + c : nullptr; });
    }
    // natural rows: tile t row r <-> Wm row 32 t + r (zero beyond out_dim)
    template <class ColFn>
    void layer(const float* Wm, int out_dim, int in_dim, int tiles, int ksteps, bool pad_end, ColFn col) {
        layer_rc(Wm, in_dim, tiles, ksteps, pad_end, [out_dim](int t, int r) { return 32 * t + r < out_dim ? 32 * t + r : -1; }, col);
    }
    // the TRANSPOSE of columns [col0, col0 + nrow) of Wm [out_dim, in_dim]: row r of the layer = input col0 + r, column c = output c;
    // columns >= out_dim come from `extra` [*, nrow] (a second matrix reading the same input, appended along the outputs)
    template <class ColFn>
    void layer_transposed(const float* Wm, int out_dim, int in_dim, int col0, int nrow, const float* extra, int extra_out, int tiles, int ksteps,
                          bool pad_end, ColFn col) {
        layer_at(tiles, ksteps, pad_end, [nrow](int t, int r) { return 32 * t + r < nrow ? 32 * t + r : -1; }, col,
                 [=](int r, int c) -> const float* {
                     if (c < out_dim) return Wm + (size_t)c * in_dim + col0 + r;
                     return c - out_dim < extra_out ? extra + (size_t)(c - out_dim) * nrow + r : nullptr;
                 });
    }
    void pad() {
        bytes.resize(cdiv((long)bytes.size(), (long)cb) * cb, 0);
        if (arena) src.resize(bytes.size() / frag_bytes(prec) * 512, -1);
    }
};

// ---------------------------------------------------------------------------------------------------------------------------
// Compensated float16 mode (EVD_PREC_F16C, mlp_pipe_c.h): every float16 fragment stream is accompanied by block-scaled fp6 (e2m3)
// fragments of the weights' rounding residual  Wl = W - f16(W)  and of the weights themselves, consumed by
// v_mfma_scale_f32_32x32x64_f8f6f4 against fp6 copies of the activations resp. of THEIR rounding residuals:
//     W x  ~=  f16(W) f16(x)  +  fp6(Wl) fp6(f16(x))  +  fp6(W) fp6(x - f16(x))
// One power-of-two scale per weight row and layer for each of the two fp6 operands (e8m0 bytes, `scales`).

// magnitude code (5 bits) of |v| on the e2m3 grid {0, 1/8 .. 7/8, 1 .. 7.5}: round to nearest even, saturate at 7.5
__host__ __device__ static inline int e2m3_mag(float a) {
    if (!(a < 7.5f)) return 31;
    if (a < 1.f) return (int)rintf(a * 8.f);                 // subnormals, step 1/8; 8 = the code of 1.0
    int e = a < 2.f ? 1 : (a < 4.f ? 2 : 3);
    const float step = (e == 1 ? 0.125f : (e == 2 ? 0.25f : 0.5f));
    const int q = (int)rintf(a / step);                       // 8 .. 16
    return q == 16 ? ((e + 1) << 3) : ((e << 3) | (q - 8));
}
__host__ __device__ static inline int e2m3_code(float v) { return e2m3_mag(v < 0.f ? -v : v) | (v < 0.f ? 32 : 0); }
__host__ __device__ static inline float e2m3_value(int code) {
    const int e = (code >> 3) & 3, m = code & 7;
    const float a = e == 0 ? m * 0.125f : (8 + m) * (e == 1 ? 0.125f : (e == 2 ? 0.25f : 0.5f));
    return (code & 32) ? -a : a;
}
// e8m0 byte of the smallest power of two s with amax / s <= 7.5
__host__ __device__ static inline int e8m0_for_max(float amax) {
    if (!(amax > 0.f)) return 64;
    int x;
    const float m = frexpf(amax, &x);                         // amax = m 2^x, m in [0.5, 1)
    int e = x - 3 + (m > 0.9375f ? 1 : 0);                    // 7.5 * 2^(x-3) = 0.9375 * 2^x
    e += 127;
    return e < 1 ? 1 : (e > 254 ? 254 : e);
}
__host__ __device__ static inline float e8m0_value(int byte) { return ldexpf(1.f, byte - 127); }
// position i (0..31) of an fp6 operand of lane half h  ->  (k-step inside the 4-k-step block, position kk = 8 h + e of that k-step)
//   both kinds are sequential: the block's elements are drained as pairs across the two source tiles (c_hid_col below), which is the order
//   v_cvt_scalef32_2xpk16_fp6_f32 interleaves its two float32 sources in, and v_cvt_scalef32_pk32_fp6_f16 reads the float16 fragments in
__host__ __device__ constexpr int c_pos_kstep(int kind, int i) { (void)kind; return i >> 3; }
__host__ __device__ constexpr int c_pos_elem(int kind, int i) { (void)kind; return i & 7; }
// input channel of a HIDDEN block's position (k-step j, position kk = 8 h + e) in this mode: the block is drained from a group of two
// output tiles in pairs (tile 0 value v, tile 1 value v) (mlp_pipe_c.h c_drain_pair), so element i = 8 (j & 3) + e is value i / 2 of tile
// i & 1, and value v of lane half h is row (v & 3) + 8 (v / 4) + 4 h of its 32-row tile (the MFMA's D layout)
__host__ __device__ constexpr int c_hid_col(int j, int kk) {
    const int b = j >> 2, e = kk & 7, h = kk >> 3, i = 8 * (j & 3) + e, v = i >> 1;
    return 64 * b + 32 * (i & 1) + (v & 3) + 8 * (v >> 2) + 4 * h;
}
// (every channel of a 256-wide layer is hit exactly once by the 16 k-steps x 16 positions)
constexpr bool c_hid_col_is_permutation() {
    bool seen[256] = {};
    for (int j = 0; j < 16; ++j)
        for (int kk = 0; kk < 16; ++kk) {
            const int c = c_hid_col(j, kk);
            if (c < 0 || c >= 256 || seen[c]) return false;
            seen[c] = true;
        }
    return true;
}
static_assert(c_hid_col_is_permutation(), "c_hid_col");

// one lane's 32 values of an fp6 operand (kind 0: the float16 rounding residuals of the weights, kind 1: the weights) -> 24 packed bytes,
// the first 16 at lo16, the last 8 at hi8; shared by the host packer and the device re-packer
__host__ __device__ static inline void c_pack_operand(int kind, const float* w, int scale_byte, uint8_t* lo16, uint8_t* hi8) {
    const float s = e8m0_value(scale_byte);
    uint32_t words[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 32; ++i) {
        const float v = kind == 0 ? w[i] - (float)(_Float16)w[i] : w[i];
        const uint32_t code = (uint32_t)e2m3_code(v / s);
        const int bit = 6 * i;
        words[bit >> 5] |= code << (bit & 31);
        if ((bit & 31) > 26) words[(bit >> 5) + 1] |= code >> (32 - (bit & 31));
    }
    for (int q = 0; q < 4; ++q) reinterpret_cast<uint32_t*>(lo16)[q] = words[q];
    reinterpret_cast<uint32_t*>(hi8)[0] = words[4];
    reinterpret_cast<uint32_t*>(hi8)[1] = words[5];
}

struct StreamBuilderC {
    size_t cb;
    std::vector<uint8_t> bytes;
    std::vector<uint32_t> scales;   // per output tile (stream order) 32 words: byte 0 = scale of the Wl operand of that row, byte 1 = of the W operand
    // for the device re-pack (repack_stream_c): where every element comes from in the parameter arena (-1: zero) and where it goes
    const float* arena = nullptr;
    std::vector<int32_t> main_src;  // [float16 fragment][lane][8]
    std::vector<uint32_t> main_dst; // byte offset of each float16 fragment
    std::vector<int32_t> op_src;    // [fp6 operand][lane][32]
    std::vector<uint32_t> op_meta;  // per fp6 operand: byte offset of its 16-byte parts, of its 8-byte parts, first scale word of its tile, kind
    explicit StreamBuilderC(size_t chunk) : cb(chunk) {}
    // One layer: groups of G tiles; per group, per block of <= 4 k-steps:  [float16 fragments: k-step major, tile minor]
    // [first 16 bytes per lane of the fp6 operands: kind major, tile minor][last 8 bytes per lane: kind major, tile minor]; zero-padded
    // to a chunk boundary at the end.  row(tile, r) -> source row or -1; col(j, kk) -> source column or -1; at(row, col) -> address or null.
    template <class RowFn, class ColFn, class AtFn>
    void layer_at(int tiles, int ksteps, int G, RowFn row, ColFn col, AtFn at) {
        auto wp = [&](int tile, int r, int j, int kk) -> const float* {
            if (j >= ksteps) return nullptr;
            const int rr = row(tile, r), c = col(j, kk);
            return (rr >= 0 && c >= 0) ? at(rr, c) : nullptr;
        };
        auto wt = [&](int tile, int r, int j, int kk) -> float { const float* w = wp(tile, r, j, kk); return w ? *w : 0.f; };
        auto idx = [&](const float* w) -> int32_t { return (w && arena) ? (int32_t)(w - arena) : -1; };
        const int nblk = (ksteps + 3) / 4;
        const uint32_t scale0 = (uint32_t)scales.size();       // first scale word of this layer
        // row scales over the whole layer
        std::vector<int> sc((size_t)tiles * 32 * 2);
        for (int tile = 0; tile < tiles; ++tile)
            for (int r = 0; r < 32; ++r) {
                float ml = 0.f, mh = 0.f;
                for (int j = 0; j < ksteps; ++j)
                    for (int kk = 0; kk < 16; ++kk) {
                        const float w = wt(tile, r, j, kk), wl = w - (float)(_Float16)w;
                        ml = fmaxf(ml, fabsf(wl));
                        mh = fmaxf(mh, fabsf(w));
                    }
                sc[(tile * 32 + r) * 2] = e8m0_for_max(ml);
                sc[(tile * 32 + r) * 2 + 1] = e8m0_for_max(mh);
                scales.push_back((uint32_t)sc[(tile * 32 + r) * 2] | ((uint32_t)sc[(tile * 32 + r) * 2 + 1] << 8));
            }
        for (int p = 0; p < tiles / G; ++p)
            for (int b = 0; b < nblk; ++b) {
                const int nk = ksteps - 4 * b < 4 ? ksteps - 4 * b : 4;
                for (int jj = 0; jj < nk; ++jj)
                    for (int t = 0; t < G; ++t) {
                        const size_t base = bytes.size();
                        bytes.resize(base + 1024, 0);
                        main_dst.push_back((uint32_t)base);
                        for (int l = 0; l < 64; ++l)
                            for (int e = 0; e < 8; ++e) {
                                *reinterpret_cast<_Float16*>(bytes.data() + base + l * 16 + e * 2) = (_Float16)wt(p * G + t, l & 31, 4 * b + jj, 8 * (l >> 5) + e);
                                main_src.push_back(idx(wp(p * G + t, l & 31, 4 * b + jj, 8 * (l >> 5) + e)));
                            }
                    }
                const size_t lo = bytes.size(), hi = lo + (size_t)2 * G * 1024;
                bytes.resize(hi + (size_t)2 * G * 512, 0);
                for (int kind = 0; kind < 2; ++kind)
                    for (int t = 0; t < G; ++t) {
                        const int tile = p * G + t;
                        op_meta.push_back((uint32_t)(lo + (size_t)(kind * G + t) * 1024));
                        op_meta.push_back((uint32_t)(hi + (size_t)(kind * G + t) * 512));
                        op_meta.push_back(scale0 + (uint32_t)tile * 32);
                        op_meta.push_back((uint32_t)kind);
                        for (int l = 0; l < 64; ++l) {
                            const int r = l & 31, h = l >> 5;
                            float v[32];
                            for (int i = 0; i < 32; ++i) {
                                const float* w = wp(tile, r, 4 * b + c_pos_kstep(kind, i), 8 * h + c_pos_elem(kind, i));
                                v[i] = w ? *w : 0.f;
                                op_src.push_back(idx(w));
                            }
                            c_pack_operand(kind, v, sc[(tile * 32 + r) * 2 + kind], bytes.data() + lo + (size_t)(kind * G + t) * 1024 + l * 16,
                                           bytes.data() + hi + (size_t)(kind * G + t) * 512 + l * 8);
                        }
                    }
            }
        bytes.resize(cdiv((long)bytes.size(), (long)cb) * cb, 0);
    }
    template <class ColFn>
    void layer(const float* Wm, int out_dim, int in_dim, int tiles, int ksteps, int G, ColFn col) {
        layer_at(tiles, ksteps, G, [out_dim](int t, int r) { return 32 * t + r < out_dim ? 32 * t + r : -1; }, col,
                 [=](int r, int c) { return c < in_dim ? Wm + (size_t)r * in_dim + c : nullptr; });
    }
};

// the compensated-float16 stream on the device with everything its re-pack from new parameter values needs
struct PackedStreamC {
    DevBuf data, scales, main_src, main_dst, op_src, op_meta, rowmax;
    long nmain = 0, nops = 0, nrows = 0;
    void release() { data.release(); scales.release(); main_src.release(); main_dst.release(); op_src.release(); op_meta.release(); rowmax.release(); }
    int upload(const StreamBuilderC& sb) {
        nmain = (long)sb.main_dst.size(); nops = (long)sb.op_meta.size() / 4; nrows = (long)sb.scales.size();
        int rc = data.upload(sb.bytes.data(), sb.bytes.size());
        if (!rc) rc = scales.upload(sb.scales.data(), sb.scales.size() * sizeof(uint32_t));
        if (!rc) rc = main_src.upload(sb.main_src.data(), sb.main_src.size() * sizeof(int32_t));
        if (!rc) rc = main_dst.upload(sb.main_dst.data(), sb.main_dst.size() * sizeof(uint32_t));
        if (!rc) rc = op_src.upload(sb.op_src.data(), sb.op_src.size() * sizeof(int32_t));
        if (!rc) rc = op_meta.upload(sb.op_meta.data(), sb.op_meta.size() * sizeof(uint32_t));
        if (!rc) rc = rowmax.alloc((size_t)nrows * 2 * sizeof(float));
        return rc;
    }
};

static __global__ void k_c_main(const float* __restrict__ arena, const int* __restrict__ src, const uint32_t* __restrict__ fdst, long n, uint8_t* __restrict__ dst) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int s = src[i];
    *reinterpret_cast<_Float16*>(dst + fdst[i >> 9] + ((i >> 3) & 63) * 16 + (i & 7) * 2) = (_Float16)(s < 0 ? 0.f : arena[s]);
}
// thread = (operand, lane): maxima of the lane's 32 values into the row's slot (non-negative floats order like their bit patterns)
static __global__ void k_c_rowmax(const float* __restrict__ arena, const int* __restrict__ src, const uint32_t* __restrict__ meta, long n, unsigned* __restrict__ rowmax) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long op = i >> 6;
    const int lane = (int)(i & 63), kind = (int)meta[op * 4 + 3];
    float m = 0.f;
    for (int e = 0; e < 32; ++e) {
        const int s = src[i * 32 + e];
        const float w = s < 0 ? 0.f : arena[s];
        m = fmaxf(m, fabsf(kind == 0 ? w - (float)(_Float16)w : w));
    }
    atomicMax(rowmax + ((long)meta[op * 4 + 2] + (lane & 31)) * 2 + kind, __float_as_uint(m));
}
static __global__ void k_c_scales(const unsigned* __restrict__ rowmax, long nrows, uint32_t* __restrict__ scales) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < nrows) scales[i] = (uint32_t)e8m0_for_max(__uint_as_float(rowmax[2 * i])) | ((uint32_t)e8m0_for_max(__uint_as_float(rowmax[2 * i + 1])) << 8);
}
static __global__ void k_c_ops(const float* __restrict__ arena, const int* __restrict__ src, const uint32_t* __restrict__ meta, const uint32_t* __restrict__ scales, long n,
                               uint8_t* __restrict__ dst) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long op = i >> 6;
    const int lane = (int)(i & 63), kind = (int)meta[op * 4 + 3];
    float w[32];
    for (int e = 0; e < 32; ++e) {
        const int s = src[i * 32 + e];
        w[e] = s < 0 ? 0.f : arena[s];
    }
    const int sb = (int)((scales[meta[op * 4 + 2] + (lane & 31)] >> (8 * kind)) & 255u);
    c_pack_operand(kind, w, sb, dst + meta[op * 4] + lane * 16, dst + meta[op * 4 + 1] + lane * 8);
}
// re-pack the compensated-float16 stream and its row scales on the device from new parameter values (stream-ordered, no host sync)
static inline int repack_stream_c(PackedStreamC& s, const float* params, hipStream_t st) {
    if (!s.data.p) return EVD_OK;
    EVD_HIP(hipMemsetAsync(s.rowmax.p, 0, s.rowmax.bytes, st));
    const long nm = s.nmain * 512, no = s.nops * 64;
    hipLaunchKernelGGL(k_c_main, dim3((unsigned)cdiv(nm, 256L)), dim3(256), 0, st, params, (const int*)s.main_src.p, (const uint32_t*)s.main_dst.p, nm, (uint8_t*)s.data.p);
    hipLaunchKernelGGL(k_c_rowmax, dim3((unsigned)cdiv(no, 256L)), dim3(256), 0, st, params, (const int*)s.op_src.p, (const uint32_t*)s.op_meta.p, no, (unsigned*)s.rowmax.p);
    hipLaunchKernelGGL(k_c_scales, dim3((unsigned)cdiv(s.nrows, 256L)), dim3(256), 0, st, (const unsigned*)s.rowmax.p, s.nrows, (uint32_t*)s.scales.p);
    hipLaunchKernelGGL(k_c_ops, dim3((unsigned)cdiv(no, 256L)), dim3(256), 0, st, params, (const int*)s.op_src.p, (const uint32_t*)s.op_meta.p, (const uint32_t*)s.scales.p, no,
                       (uint8_t*)s.data.p);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

// a packed fragment stream on the device and, per element, the index of its source in the parameter arena (-1: zero)
struct PackedStream {
    int prec = 0;
    DevBuf data, src;
    void release() { data.release(); src.release(); }
    int upload(const StreamBuilder& sb) {
        prec = sb.prec;
        int rc = data.upload(sb.bytes.data(), sb.bytes.size());
        if (!rc) rc = src.upload(sb.src.data(), sb.src.size() * sizeof(int32_t));
        return rc;
    }
};

static __global__ void k_pack_stream(int prec, const float* __restrict__ arena, const int* __restrict__ src, long n, uint8_t* __restrict__ dst) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int s = src[i];
    put_element(prec, dst + (i >> 9) * frag_bytes(prec), (int)(i >> 3) & 63, (int)i & 7, s < 0 ? 0.f : arena[s]);
}

static __global__ void k_gather_f32(const float* __restrict__ arena, const int* __restrict__ src, long n, float* __restrict__ dst) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i] < 0 ? 0.f : arena[src[i]];
}

// re-pack one stream on the device from new parameter values
static inline int repack_stream(PackedStream& s, const float* params, hipStream_t st) {
    if (!s.data.p) return EVD_OK;
    const long nel = (long)(s.src.bytes / sizeof(int32_t));
    if (!nel) return EVD_OK;
    hipLaunchKernelGGL(k_pack_stream, dim3((unsigned)cdiv(nel, 256L)), dim3(256), 0, st, s.prec, params, (const int*)s.src.p, nel, (uint8_t*)s.data.p);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

// ---- every stream of a handle re-packed by ONE launch (a training iteration re-packs ~25 streams per level after optimizer.step():
// as separate 4 us launches they were 2.3 % of a whole blurfactory iteration)
struct RepackSeg { const int* src; uint8_t* dst; long nel; int prec, pad; };
struct RepackBatch {
    DevBuf table;
    int nseg = 0;
    long max_nel = 0;
    void release() { table.release(); nseg = 0; max_nel = 0; }
    int build(const std::vector<PackedStream*>& list) {
        std::vector<RepackSeg> segs;
        for (PackedStream* s : list) {
            const long nel = s->data.p ? (long)(s->src.bytes / sizeof(int32_t)) : 0;
            if (!nel) continue;
            segs.push_back(RepackSeg{(const int*)s->src.p, (uint8_t*)s->data.p, nel, s->prec, 0});
            max_nel = nel > max_nel ? nel : max_nel;
        }
        nseg = (int)segs.size();
        return nseg ? table.upload(segs.data(), segs.size() * sizeof(RepackSeg)) : EVD_OK;
    }
};

static __global__ __launch_bounds__(256) void k_pack_streams(const RepackSeg* __restrict__ segs, const float* __restrict__ arena) {
    const RepackSeg s = segs[blockIdx.y];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < s.nel; i += (long)gridDim.x * 256) {
        const int k = s.src[i];
        put_element(s.prec, s.dst + (i >> 9) * frag_bytes(s.prec), (int)(i >> 3) & 63, (int)i & 7, k < 0 ? 0.f : arena[k]);
    }
}

static inline int repack_batch(RepackBatch& b, const std::vector<PackedStream*>& list, const float* params, hipStream_t st) {
    if (!b.nseg && !b.table.p) {
        int rc = b.build(list);
        if (rc) return rc;
    }
    if (!b.nseg) return EVD_OK;
    const long bx = cdiv(b.max_nel, 256L) < 512 ? cdiv(b.max_nel, 256L) : 512;
    hipLaunchKernelGGL(k_pack_streams, dim3((unsigned)bx, (unsigned)b.nseg), dim3(256), 0, st, (const RepackSeg*)b.table.p, params);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

}  // namespace evd
