// Host-side packing of nn.Linear weights into MFMA fragment streams (contract: nerf_mlp.h).
#pragma once

#include <cstdint>
#include <cstring>
#include <vector>

#include "evd_common.h"
#include "nerf_mlp.h"

namespace evd {

static inline uint16_t f32_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN
    u += 0x7fffu + ((u >> 16) & 1u);                                            // round to nearest even
    return (uint16_t)(u >> 16);
}

struct StreamBuilder {
    int prec;
    int group = 2;                  // tiles per group (kernel's G); 1-tile layers are always their own group
    size_t cb;                      // chunk bytes: layers flagged pad_end are zero-padded to a multiple of it
    std::vector<uint8_t> bytes;
    explicit StreamBuilder(int p, size_t chunk = 0) : prec(p), cb(chunk ? chunk : (size_t)chunk_bytes(p)) {}
    // one fragment of output tile `tile`, k-step j.  row(tile, r) -> source row of Wm or -1 (zero row);
    // col(j, kk) -> source column or -1 (zero padding)
    template <class RowFn, class ColFn>
    void frag(const float* Wm, int in_dim, int tile, int j, RowFn row, ColFn col) {
        const int fb = frag_bytes(prec);
        const size_t base = bytes.size();
        bytes.resize(base + fb, 0);
        uint8_t* dst = bytes.data() + base;
        for (int l = 0; l < 64; ++l) {
            const int r = row(tile, l & 31);
            for (int e = 0; e < 8; ++e) {
                const int kk = 8 * (l >> 5) + e;
                const int c = col(j, kk);
                const float w = (r >= 0 && c >= 0 && c < in_dim) ? Wm[(size_t)r * in_dim + c] : 0.f;
                if (prec == EVD_PREC_BF16) {
                    const uint16_t b = f32_to_bf16(w);
                    memcpy(dst + l * 16 + e * 2, &b, 2);
                } else if (prec == EVD_PREC_F16) {
                    const _Float16 hv = (_Float16)w;
                    memcpy(dst + l * 16 + e * 2, &hv, 2);
                } else if (prec == EVD_PREC_F16X3) {
                    const _Float16 hi = (_Float16)w;
                    const _Float16 lo = (_Float16)((w - (float)hi) * 2048.f);
                    memcpy(dst + l * 16 + e * 2, &hi, 2);
                    memcpy(dst + 1024 + l * 16 + e * 2, &lo, 2);
                } else {
                    memcpy(dst + (e < 4 ? 0 : 1024) + l * 16 + (e & 3) * 4, &w, 4);
                }
            }
        }
    }
    // fragments in kernel order: tile groups of 2 (or 1), k-steps inside, tiles of the group innermost
    template <class RowFn, class ColFn>
    void layer_rc(const float* Wm, int in_dim, int tiles, int ksteps, bool pad_end, RowFn row, ColFn col) {
        const int G = (tiles % group == 0) ? group : 1;
        for (int p = 0; p < tiles / G; ++p)
            for (int j = 0; j < ksteps; ++j)
                for (int t = 0; t < G; ++t) frag(Wm, in_dim, p * G + t, j, row, col);
        if (pad_end) pad();
    }
    // natural rows: tile t row r <-> Wm row 32 t + r (zero beyond out_dim)
    template <class ColFn>
    void layer(const float* Wm, int out_dim, int in_dim, int tiles, int ksteps, bool pad_end, ColFn col) {
        layer_rc(Wm, in_dim, tiles, ksteps, pad_end, [out_dim](int t, int r) { return 32 * t + r < out_dim ? 32 * t + r : -1; }, col);
    }
    void pad() {
        bytes.resize(cdiv((long)bytes.size(), (long)cb) * cb, 0);
    }
};

}  // namespace evd
