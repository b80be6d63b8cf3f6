// Host-side packing of nn.Linear weights into MFMA fragment streams (contract: nerf_mlp.h).
#pragma once

#include <cstdint>
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
    hipLaunchKernelGGL(k_pack_stream, dim3((unsigned)cdiv(nel, 256L)), dim3(256), 0, st, s.prec, params, (const int*)s.src.p, nel, (uint8_t*)s.data.p);
    EVD_LAUNCH_CHECK();
    return EVD_OK;
}

}  // namespace evd
