// Shared between kernel_voxel.hip (kernels + launchers) and evd_voxel_api.hip (C ABI).
#pragma once

#include "evd_common.h"

namespace evd {

struct GridParams {
    const float* plane[3];      // channel-last [grid[m1]][grid[m0]][C_i]
    const float* line[3];       // [grid[vec]][C_i]
    const _Float16* plane_h[3]; // float16 copies of the grids (same layout), read by the bf16 / f16 arithmetic modes:
    const _Float16* line_h[3];  //   half the gather bytes; 2^-11 rounding, the same as those modes' MFMA operands
    const float* basis;         // [app_dim][Ctot]
    int n_comp[3], grid[3], app_dim, app_act;
    float aabb_min[3], inv[3];  // inv = 2 / (max - min)  (invaabbSize, voxnerf.py:91)
};

struct GridGrads { float *plane[3], *line[3], *basis; };    // channel-last like the grids; null = not wanted

struct VoxMlpParams {
    const char* wstream;
    const float* bias;          // 512 zero floats (bias-free sigma net) then the colour-net biases, 32 per output tile
    const float* pts;           // [n,3]
    const float* viewdirs;      // rows of vd_stride floats, one per ray
    const float* fts;           // [n, ft_stride]
    long nsamp;
    int S, vd_stride, ft_stride, nchunks, nbias;
    float* raw;                 // [n,4] = (sigma, sigmoid(colour))  voxnerf.py:254
    float* feature;             // [n,G] or null
    char* act;                  // training kernels: activation store, else null
    int pe_l = 10, pe_lv = 4;           // multires / multires_views (generic kernel; the pipelined ones are built for nerf_mlp.h PE_L / PE_LV)
    const unsigned* wscale = nullptr;   // compensated float16 mode: row-scale words, 32 per output tile in stream order (pack.h StreamBuilderC)
    int rev_trig = 0;                   // k_voxel_mlp_resident: the encodings' sines on the hardware unit behind the two-float revolution reduction (the f16c render's coarse level)
};

// ---- binned scatter (kernel_voxel_scatter.hip): what the first pass of the tri-plane backward leaves per sample for the second
struct PTap { unsigned short cx0, cx1, cy0, cy1; float w[4]; };     // clamped cell coordinates of the 4 bilinear taps of one plane + weights (0 = outside)
struct LTap { int c0, c1; float w0, w1; };                          // the 2 taps of one line
constexpr int SC_TS = 16;                                           // plane tiles of 16 x 16 cells (+ 1 halo row / column in LDS)
struct BinOut {
    float* rows_p;          // [n, ctot]  d coef x line value: what every plane tap adds, times its weight ([n, n_comp[0]]: the x-y plane only, in the hybrid form)
    float* rows_l;          // [n, ctot]  d coef x plane value, for the line taps
    PTap* ptap;             // [n, 3]  ([n]: the x-y plane only, in the hybrid form)
    LTap* ltap;             // [n, 3]
    unsigned* keys[3];      // [n] tile of tap 0 in plane i
    unsigned* ids;          // [n] 0 .. n-1 (the sort's values)
    int tiles_x[3];
};
bool voxel_scatter_binned_ok(const GridParams& g, const GridGrads& gg, long n);
size_t voxel_scatter_workspace_bytes(const GridParams& g, long n);
int launch_voxel_sample_bwd_pass1(const GridParams& g, const float* pts, long n, const float* d_out, int d_stride, int d_col, const GridGrads& gg,
                                  float* d_pts, const BinOut& bo, hipStream_t st);
int launch_voxel_sample_bwd_planes(const GridParams& g, const float* pts, long n, const float* d_out, int d_stride, int d_col, const GridGrads& gg,
                                   float* d_pts, const BinOut& bo, hipStream_t st);
bool voxel_sample_bwd_w_ok(const GridParams& g);
bool voxel_sample_bwd_w_lines12(const GridParams& g);
int launch_voxel_sample_bwd_w(const GridParams& g, const float* pts, long n, const float* d_out, int d_stride, int d_col, const GridGrads& gg,
                              float* d_pts, float* rows_l, LTap* ltap, float* coef, hipStream_t st, unsigned* lmax = nullptr, bool half_grids = false);
size_t voxel_scatter_hybrid_workspace_bytes(const GridParams& g, long n);
// half_grids: the re-gather of the grid values reads the float16 copies (the forward of the half-precision modes did: kernel_voxel.hip HALF)
int launch_voxel_sample_bwd_hybrid(const GridParams& g, const float* pts, long n, const float* d_out, int d_stride, int d_col, const GridGrads& gg,
                                   float* d_pts, void* workspace, size_t workspace_bytes, hipStream_t st, bool half_grids = false);
int launch_voxel_sample_bwd_binned(const GridParams& g, const float* pts, long n, const float* d_out, int d_stride, int d_col, const GridGrads& gg,
                                   float* d_pts, void* workspace, size_t workspace_bytes, hipStream_t st);

constexpr int TV_MAX_BLOCKS = 4096;     // partial (dh^2, dw^2) pairs per tensor
struct TvShape { int C[6], H[6], W[6], blocks[6]; };
// one launch for the six tensors of a level (round 6: twelve launches per iteration -- half of them on 586 x 64 line grids, whose kernels are
// all launch -- became two, for the value and for the gradient): the launch's blocks are dealt to the jobs in order
struct TvJob { const float* x; float* grad; int H, W, C; float weight; int blk0, nblk; };
struct TvJobs { TvJob j[6]; int n; };
int launch_tv_level(TvJobs& jobs, double* partials /* job i at i * 2 * TV_MAX_BLOCKS */, TvShape* shape, hipStream_t st);
int launch_tv_bwd_level(TvJobs& jobs, const float* d_loss, hipStream_t st);

int launch_points(const float* rb, int nc, const float* z, long n, int S, float* pts, hipStream_t st);
// evd_sample_z + the sample positions (kernels_render.hip)
int launch_sample_z_pts(const evd_render_cfg* cfg, const float* ray_batch, int ncol, long R, const float* t_rand, float* z, float* pts, hipStream_t stream);
// evd_sample_pdf_merge + the positions o + d z of the new / of the merged samples (kernels_render.hip)
int launch_sample_pdf_merge_pts(const float* z, const float* weights, long R, int S, int N, int det, const float* u,
                                float* z_samples, float* z_merged, int* order, float* z_std,
                                const float* rb, int rb_cols, float* pts_new, float* pts_merged, hipStream_t stream);
int launch_merge_features(const float* old, const float* fresh, const int* order, long R, int S, int N, int F, float* out, int out_stride,
                          hipStream_t st);
int launch_merge_features_bwd(const float* d_out, int d_stride, const int* order, long R, int S, int N, int F, float* d_old, float* d_fresh, hipStream_t st);
int launch_voxel_sample(const GridParams& g, bool half_grids, const float* pts, long n, float* out, int out_stride, int out_col, hipStream_t st);
int launch_voxel_sample_bwd(const GridParams& g, const float* pts, long n, const float* d_out, int d_stride, int d_col, const GridGrads& gg,
                            float* d_pts, hipStream_t st);
int launch_tv_bwd(const float* x, int H, int W, int C, const float* d_loss, float weight, float* grad, hipStream_t st);
int launch_f32_to_f16(const float* x, long n, _Float16* y, hipStream_t st);
int launch_tv(const float* x, int H, int W, int C, double* partials, int* blocks, hipStream_t st);
int launch_tv_finish(const double* acc, const TvShape& s, float* out, hipStream_t st);
int voxel_mlp_dispatch(int prec, int HD, int G, int FT, const VoxMlpParams& p, hipStream_t st);
int voxel_mlp_c_chunks(int HD, int G, int FT);              // compensated float16 mode (voxel_mlp_c_kernel.h): chunks of its stream, 0 = not built for this level
int launch_voxel_pipe_f16c(const VoxMlpParams& p, hipStream_t st);
int launch_voxel_train_fwd_f16c(const VoxMlpParams& p, hipStream_t st);   // fine level; writes the float16 mode's activation store

}  // namespace evd
