// Launchers of the NeRF-mode training kernels (kernel_nerf_train_*.hip), called by evd_train_api.hip.
#pragma once

#include "evd_common.h"
#include "nerf_mlp.h"
#include "nerf_net.h"

namespace evd {

int launch_nerf_train_fwd_f16(const MlpParams& p, hipStream_t st);
int launch_nerf_train_fwd_bf16(const MlpParams& p, hipStream_t st);
int launch_nerf_train_fwd_f16x3(const MlpParams& p, hipStream_t st);
int launch_nerf_train_fwd_f16x3_hi(const MlpParams& p, hipStream_t st);     // EVD_PREC_F16M: split-float16 forward, the float16 mode's store
int launch_nerf_train_fwd_f16c(const MlpParams& p, hipStream_t st);         // EVD_PREC_F16C: compensated forward, the float16 mode's store

// index maps of the wgrad reduction (fragment column -> parameter row / column, -1 = padding), offsets into one int32 array
enum { MAP_HID = 0,                    // 256: hidden arrangement, channel 16 j + phi(kk)
       MAP_HID_SKIP = MAP_HID + 256,   // 256: the same behind the 63 point-encoding columns of the skip layer
       MAP_PE = MAP_HID_SKIP + 256,    // 64: point encoding (4 fragments)
       MAP_DIR = MAP_PE + 64,          // 32: direction encoding (2 fragments), behind the 256 feature columns
       MAP_RGB = MAP_DIR + 32,         // 32: d rgb fragment
       MAP_ALPHA = MAP_RGB + 32,       // 32: d alpha fragment
       MAP_TOTAL = MAP_ALPHA + 32 };

struct BwdGrads {                       // device float32, reference nn.Linear layouts; null = not wanted
    float *pts_w[EVD_MAX_LAYERS], *pts_b[EVD_MAX_LAYERS];
    float *views_w, *views_b, *feature_w, *feature_b, *alpha_w, *alpha_b, *rgb_w, *rgb_b;
};

struct BwdPlan {
    const float* d_raw;                 // [nsamp, 4]
    long nsamp, tiles;
    char* store;
    const char* wt[EVD_BWD_NSTREAMS];   // W^T fragment streams
    const int* maps;
    float* partial;
    unsigned* maxbits;
    int wgrad_blocks, skip;
    hipStream_t side;                   // second stream for the wgrad launches (null: everything on the caller's stream)
    hipEvent_t ev;
    BwdGrads grads;
    int accumulate = 0;                 // 1: the parameter gradients are ADDED into the caller's buffers (evd_nerf_grads.accumulate)
    const float *pts, *viewdirs;        // [nsamp,3] sample positions / rows of vd_stride floats per ray (the encodings' derivatives)
    int vd_stride, S;
    float *d_pts, *d_dirs;              // [nsamp,3] float32 out (through the positional encodings), or null
};

int run_nerf_backward_f16(const BwdPlan& b, hipStream_t st);
int run_nerf_backward_bf16(const BwdPlan& b, hipStream_t st);
int run_nerf_backward_f16x3(const BwdPlan& b, hipStream_t st);

}  // namespace evd
