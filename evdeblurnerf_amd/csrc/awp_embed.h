// AWP consumer: AdaptiveWeightProposal.sample_feature_embed_layer, reference networks/dpnerf/awp.py:36-37,98-100
// (D_sam = 4 x [Linear + ReLU], 128 -> 64 -> 64 -> 64 -> 64: the shipped kernel_awp_sam_emb_depth / _width and fine_geo_feat_dim),
// on the software pipeline of mlp_pipe.h.  Its input is the fine PDRF level's per-sample geo features; in training they are read as
// the float16 / bfloat16 MFMA fragments the level's training forward already keeps for its own backward (voxel_mlp_kernel.h
// VStore::GEO), so the reference's depth_feature tensor [R P, S, 128] (renderer.py:253-256,314) is never written.
// Shared between the kernels (awp_embed_kernel.h) and the C ABI (evd_awp_api.hip).
#pragma once

#include "evd_common.h"

namespace evd {

constexpr int AWP_IN = 128, AWP_W = 64, AWP_D = 4;

// fragment slots of the embed network's own activation / gradient store, per 32-sample tile (1 KiB fragments, lane-linear like
// nerf_mlp.h astore).  The geo fragments are copied in by the forward so that the store is self-contained for wgrad.
namespace awpstore {
constexpr int GEO = 0, E0 = GEO + AWP_IN / 16;                              // E_l at E0 + 4 l: the ReLU output of layer l
constexpr int FWD_END = E0 + AWP_D * (AWP_W / 16);
constexpr int D_E0 = FWD_END;                                              // d loss / d pre-activation of layer l at D_E0 + 4 l
constexpr int D_GEO = D_E0 + AWP_D * (AWP_W / 16);                         // d geo, handed to the fine level's backward
constexpr int TILE_FRAGS = D_GEO + AWP_IN / 16;
constexpr long TILE_BYTES = (long)TILE_FRAGS * 1024;
constexpr int TRAILER_BYTES = 256;     // behind the tiles: word 0 = float bits of max |d h_local| (the chain's loss scale),
                                       //                   word 1 = float bits of max |d geo| in true units
}  // namespace awpstore

inline long awp_tiles(long nsamp) { return cdiv(nsamp, 256L) * 8; }       // = evd_voxel_api.hip vox_tiles: both stores tile alike

// wgrad index maps (fragment column -> parameter row / column), offsets into one int32 array
enum { AMAP_H = 0, AMAP_GEO = AMAP_H + AWP_W, AMAP_TOTAL = AMAP_GEO + AWP_IN };

struct AwpFwdParams {
    const char* wstream;
    const float* bias;          // AWP_D x AWP_W floats, layer order
    const float* geo_rows;      // [n, 128] float32, or null: read the fragments
    const char* geo_frags;      // the fine level's activation store (tiles of geo_tile_bytes, fragments from geo_slot), or null
    long geo_tile_bytes;
    int geo_slot;
    long nsamp;
    float* h_local;             // [n, 64] float32 out
    char* act;                  // own store (training) or null
    int nchunks;
};

struct AwpBwdGrads { float *w[AWP_D], *b[AWP_D]; };     // device float32, nn.Linear layouts; null = not wanted

struct AwpBwdPlan {
    const float* d_h_local;     // [n, 64]
    const unsigned* d_h_absmax; // float bits of max |d_h_local| when the producer took it, or null
    long nsamp, tiles;
    char* store;
    const char* wt[AWP_D];      // W_l^T fragment streams
    const int* maps;
    float* partial;
    int wgrad_blocks;
    float* d_geo_rows;          // [n, 128] float32 out, or null
    AwpBwdGrads grads;
};

int launch_awp_embed_f16(bool train, const AwpFwdParams& p, hipStream_t st);
int launch_awp_embed_bf16(bool train, const AwpFwdParams& p, hipStream_t st);
int run_awp_backward_f16(const AwpBwdPlan& b, hipStream_t st);
int run_awp_backward_bf16(const AwpBwdPlan& b, hipStream_t st);
constexpr int AWP_NCHUNKS = 3;         // 16 + 3 x 8 one-KiB fragments in 16 KiB chunks

}  // namespace evd
