// Launchers / plans of the PDRF training kernels (kernel_voxel_train_*.hip), called by evd_voxel_api.hip.
#pragma once

#include "evd_common.h"
#include "voxel.h"

namespace evd {

// W^T streams of the dgrad chain
enum { VBWD_C2 = 0, VBWD_C1, VBWD_C0, VBWD_SIGGEO, VBWD_L0, VBWD_NSTREAMS };

// wgrad index maps (fragment column -> parameter row / column, -1 = padding), offsets into one int32 array
enum { VMAP_HID = 0,                    // 256: hidden arrangement 16 j + phi(kk) (-1 beyond the hidden width)
       VMAP_FTS = VMAP_HID + 256,       // 64: input features, natural arrangement 16 j + kk
       VMAP_PE = VMAP_FTS + 64,         // 64: point encoding, behind the FT feature columns
       VMAP_GEO_X = VMAP_PE + 64,       // 128: geo channels as color_net.0 columns
       VMAP_DIR = VMAP_GEO_X + 128,     // 32: direction encoding, behind the G geo columns (adjacent to them when G = 128)
       VMAP_GEO_Y = VMAP_DIR + 32,      // 128: geo channels as sigma_net.1 rows (1 + channel)
       VMAP_COL = VMAP_GEO_Y + 128,     // 32: d colour fragment
       VMAP_SIG = VMAP_COL + 32,        // 32: d sigma fragment (row 0 of sigma_net.1)
       // blocks of the fused 64-wide backward (voxel_bwd_fused64.h): 32-position blocks that pair fragments the maps above keep apart
       VMAP_F64_SG = VMAP_SIG + 32,     // 32: rows of the gradient block [d geo fragment 0 | d sigma fragment] of sigma_net.1
       VMAP_F64_C0 = VMAP_F64_SG + 32,  // 64: columns of color_net.0: [geo fragments 0, 1 | direction encoding]
       VMAP_F64_L0 = VMAP_F64_C0 + 64,  // 96: columns of sigma_net.0: [feature fragments 0, 1 | point encoding]
       VMAP_SG5 = VMAP_F64_L0 + 96,     // 160: rows of sigma_net.1 as ONE gradient of 4 + 1 row tiles: [geo rows | the d sigma fragment] (fused wgrad + dgrad)
       VMAP_TOTAL = VMAP_SG5 + 160 };

struct VoxBwdGrads {                    // device float32, reference nn.Linear layouts; null = not wanted
    float *sigma_w[2], *color_w[3], *color_b[3];
};

struct VoxBwdPlan {
    const float *d_raw, *raw;           // [nsamp, 4]
    const float* d_feature;             // [nsamp, G] gradient of the geo-feature output, or null
    const char* awp_store;              // the AWP embedding's store after ITS backward (awp_embed.h): d geo fragments to add, or null
    long awp_tile_bytes;
    int awp_slot;
    const unsigned* awp_words;          // its trailer: loss-scale word, max |d geo| in true units
    long nsamp, tiles;
    char* store;
    const char* wt[VBWD_NSTREAMS];
    const int* maps;
    float* partial;
    unsigned* maxbits;
    int wgrad_blocks;
    hipStream_t side;                   // second stream for the wgrad launches (null: none), nerf_train.h
    hipEvent_t ev;
    float* d_fts;                       // [nsamp, d_fts_stride] float32 out, or null
    int d_fts_stride;
    const float *pts, *viewdirs;        // the forward's inputs (for the encodings' derivatives)
    int vd_stride, S;
    float *d_pts, *d_dirs;              // [nsamp, 3] float32 out (through PE(pts) / PE(dirs) only), or null
    VoxBwdGrads grads;
    int accumulate = 0;                 // 1: the parameter gradients are ADDED into the caller's buffers (evd_voxel_grads.accumulate)
};

int launch_voxel_train_fwd_f16(int HD, const VoxMlpParams& p, hipStream_t st);
int launch_voxel_train_fwd_bf16(int HD, const VoxMlpParams& p, hipStream_t st);
int launch_voxel_train_fwd_f16x3(int HD, const VoxMlpParams& p, hipStream_t st);
int launch_voxel_train_fwd_f16x3_hi(int HD, const VoxMlpParams& p, hipStream_t st);   // split-float16 arithmetic, the float16 mode's store
int run_voxel_backward_f16(int HD, const VoxBwdPlan& b, hipStream_t st);
int run_voxel_backward_bf16(int HD, const VoxBwdPlan& b, hipStream_t st);
int run_voxel_backward_f16x3(int HD, const VoxBwdPlan& b, hipStream_t st);
inline int launch_voxel_train_fwd_dispatch(int prec, int HD, const VoxMlpParams& p, hipStream_t st) {
    return prec == 3 /*EVD_PREC_F16*/ ? launch_voxel_train_fwd_f16(HD, p, st)
           : prec == 2 /*EVD_PREC_BF16*/ ? launch_voxel_train_fwd_bf16(HD, p, st) : launch_voxel_train_fwd_f16x3(HD, p, st);
}
inline int run_voxel_backward_dispatch(int prec, int HD, const VoxBwdPlan& b, hipStream_t st) {
    return prec == 3 ? run_voxel_backward_f16(HD, b, st) : prec == 2 ? run_voxel_backward_bf16(HD, b, st) : run_voxel_backward_f16x3(HD, b, st);
}
int voxel_store_geo_slot(int HD);         // first of the geo fragments the training forward keeps (fine level: 8 fragments)
long voxel_store_tile_bytes(int HD);      // fine 256 / 128 / 64 or coarse 64 / 15 / 32 (kernel_voxel_train_f16.hip); half-precision modes
long voxel_store_tile_bytes_prec(int HD, int prec);

}  // namespace evd
