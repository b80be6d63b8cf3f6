/*
 * evdnerf.h -- C ABI of libevdnerf.so: the MI355X (gfx950) renderer + blur/event-loss hot path of
 * EvDeblurNeRF, written as HIP kernels.  This is the drop-in boundary: the reference has no FFI
 * layer (it is pure PyTorch), so every entry point below replaces one Python function of the reference
 * (cited as file:line, paths relative to the reference checkout) and the Python mirror in
 * evdeblurnerf_amd/ binds them with ctypes (INTEGRATION.md shows the stub a maintainer would add).
 *
 * Conventions
 *   - every pointer marked "dev" is device memory (HBM) owned by the caller; tensors are float32,
 *     contiguous, row-major; "host" pointers are plain host memory read during the call only.
 *   - every entry returns 0 (EVD_OK) or a negative EVD_E* code; evd_last_error() gives the message
 *     of the last failure on the calling thread.
 *   - kernels are enqueued on the caller's HIP stream (void* = hipStream_t); no entry synchronises,
 *     allocates in the hot path, or keeps mutable state outside the handles it is given (the only process-wide data are
 *     write-once caches of device attributes).  Scratch memory comes from the caller
 *     (evd_*_workspace_bytes).  Opaque handles own only their packed parameters.
 *   - NULL output pointers mean "not wanted".
 */
#ifndef EVDNERF_H
#define EVDNERF_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EVD_OK 0
#define EVD_E_INVALID (-1)      /* bad argument / unsupported configuration */
#define EVD_E_HIP (-2)          /* a HIP runtime call failed */
#define EVD_E_WORKSPACE (-3)    /* workspace too small */
#define EVD_E_NODEVICE (-4)     /* no gfx950 device */

#define EVD_MAX_LAYERS 16

/* arithmetic of the per-sample MLP GEMMs (accumulation is always float32) */
#define EVD_PREC_F32 0      /* v_mfma_f32_32x32x2_f32: exact float32 products (bitwise an fmaf chain) */
#define EVD_PREC_F16X3 1    /* operands split hi+lo into two float16, 3 x v_mfma_f32_32x32x16_f16: ~2^-21 products */
#define EVD_PREC_BF16 2     /* v_mfma_f32_32x32x16_bf16: throughput mode, ~2^-8 operands */
#define EVD_PREC_F16 3      /* v_mfma_f32_32x32x16_f16, one product: bf16 speed, ~2^-11 operands, float16 range */
#define EVD_PREC_F16C 4     /* compensated float16: the float16 product + two block-scaled fp6 (e2m3) products of the operands' rounding
                             * residuals on v_mfma_scale_f32_32x32x64_f8f6f4: ~2^-15 operands at 1.4x the matrix-pipe time of EVD_PREC_F16.
                             * Built for the netdepth 8, netwidth 256, skips [4] NeRF network without feature rows; EVD_E_INVALID otherwise. */
#define EVD_NUM_PREC 5      /* arithmetic modes with weight streams of their own */
/* TRAINING entries only (evd_*_mlp_train / _backward / _train_store_bytes_prec, evd_voxel_sample_prec), round 4.  The reference trains in
 * float32 (run_nerf.py:593-601); both mixed modes below keep the single-product float16 mode's activation store and run ITS backward
 * (loss-scaled float16 operands, float32 accumulation), behind a forward that holds the float32 numbers:
 *   EVD_PREC_F16C (above) as a training mode: the compensated forward where a level / network has that kernel (PDRF fine level, 8 x 256
 *     NeRF), the split-float16 forward elsewhere (PDRF coarse level).  Forward ~2^-15; ReLU patterns differ from float32 in ~5e-6 of the
 *     units, which bounds the gradients at ~sqrt(5e-6) = a few 1e-3 of their norm whatever the batch size.
 *   EVD_PREC_F16M: the split-float16 (EVD_PREC_F16X3) forward everywhere: forward ~2^-21, float32's own ReLU patterns; gradients within
 *     the float16 backward's rounding (~1e-3).  Inference entries reject it (render in EVD_PREC_F16X3). */
#define EVD_PREC_F16M 5

/* activation codes: reference networks/nerf.py:31-33, networks/pdrf/voxnerf.py:28-30 */
#define EVD_ACT_NONE 0
#define EVD_ACT_RELU 1
#define EVD_ACT_SIGMOID 2
#define EVD_ACT_EXP 3
#define EVD_ACT_SIGMOID1 4
#define EVD_ACT_SOFTPLUS 5
#define EVD_ACT_TANH 6

const char* evd_last_error(void);
int evd_version(void);
/* test support (tests/test_gpu_dist.py; no reference counterpart): the number of delay kernels the side-stream test hook launched in this
 * process (EVD_TEST_SIDE_SPIN_US=N in the environment puts an N-microsecond spin in front of every backward launch on a handle's side
 * stream; 0 when the variable is unset -- the hook is then one branch per backward entry) */
long evd_debug_side_spin_count(void);
/* number of gfx950 devices visible; <0 on error */
int evd_device_count(void);

/* ---------------------------------------------------------------- rays (reference utils/rays.py) */
/* get_rays, utils/rays.py:8-22.  K host[9], c2w host[12] -> rays_o/rays_d dev [H,W,3].  add_halfpix: the reference's keyword
 * (:8-9; HALF_PIX = 0.5 added to the pixel index when non-zero). */
int evd_get_rays(int H, int W, const float* K, const float* c2w, int add_halfpix, float* rays_o, float* rays_d, void* stream);
/* get_rays_pix, utils/rays.py:25-36.  coords dev [n,2], c2ws dev [n,3,4], K host[9].  add_halfpix 0: the coordinates are already
 * sub-pixel positions (rectified event coordinates, data/loader_events.py:290-293 passes integer_coords). */
int evd_get_rays_pix(const float* coords, const float* K, const float* c2ws, long n, int add_halfpix,
                     float* rays_o, float* rays_d, void* stream);
/* get_ndc_rays, utils/rays.py:104-145 */
int evd_ndc_rays(int H, int W, float focal, float near, const float* rays_o, const float* rays_d, long n,
                 float* out_o, float* out_d, void* stream);
/* RigidBlurringModel.rbk_warp, networks/dpnerf/blurmodel.py:51-82 (SE3Field / RigidBody of utils/rigid_warping.py): the
 * sub-exposure rays of the blur batch.  rays dev [R,3,2]; r, v dev [R,3,M] (the r_linear / v_linear outputs viewed as
 * [R,3,M], blurmodel.py:52-53) -> new_rays dev [R, M + use_origin, 3, 2] (slot 0 = the input ray when use_origin),
 * transforms dev [R, M + use_origin, 4, 4] or NULL. */
int evd_rbk_warp(const float* rays, const float* r, const float* v, long R, int M, int use_origin, float* new_rays, float* transforms,
                 void* stream);
/* The NaN / Inf guard of render_rays, networks/renderer.py:259-263 (there: isnan().any() + isinf().any() per result key = two
 * host synchronisations per key).  Here one launch: ptrs host[n_keys] (device arrays), counts host[n_keys] (floats per array),
 * n_keys <= 16 -> flags dev [n_keys] (unsigned): bit 0 = the key contains a NaN, bit 1 = an Inf.  Nothing synchronises; the caller
 * reads the words when (if) it wants the answer. */
int evd_numerics_flags(const float* const* ptrs, const long* counts, int n_keys, unsigned* flags, void* stream);
/* Embedder.forward, networks/embedding.py:88-98.  x dev [n,dim] -> out dev [n, dim*(1+2L)] */
int evd_embed(const float* x, long n, int dim, int L, float* out, void* stream);

/* ---------------------------------------------------------------- render configuration
 * arguments of NeRFAll.render / render_rays (networks/renderer.py:399-466, 129-264) */
typedef struct {
    int H, W;
    float focal;            /* K[0][0] */
    int ndc, use_viewdirs, lindisp, N_samples, N_importance, white_bkgd;
    float near, far, perturb;
    int is_train;           /* nn.Module.training: gates render_rmnearplane (nerf.py:107) */
    int precision;          /* EVD_PREC_* for the MLP GEMMs */
} evd_render_cfg;

/* NeRFAll.render ray packing, networks/renderer.py:423-446: rays dev [R,3,2] -> ray_batch dev [R,11]
 * = o(3) d(3) near far viewdir(3); viewdir = d/|d| before the NDC warp.  (8 columns when !use_viewdirs) */
int evd_ray_batch(const evd_render_cfg* cfg, const float* rays, long R, float* ray_batch, void* stream);
/* Backward of evd_ray_batch (the training branch of NeRFAll.forward, renderer.py:303-308: the loss reaches the blur kernel's warped
 * rays through the packing, viewdirs normalisation :431 and the NDC warp utils/rays.py:104-145; the reference gets it from autograd):
 * d_ray_batch dev [R,11] -> d_rays dev [R,3,2] (overwritten).  The near / far columns carry no gradient. */
int evd_ray_batch_bwd(const evd_render_cfg* cfg, const float* rays, const float* d_ray_batch, long R, float* d_rays, void* stream);
/* pts = rays_o + rays_d * z (renderer.py:180,206,235): ray_batch dev rows of ncol floats (o at 0..2, d at 3..5), z dev [R,S] -> pts dev [R,S,3];
 * its backward: d_pts dev [R,S,3] -> columns 0..2 (sum_s d_pts) and 3..5 (sum_s z d_pts) of d_ray_batch dev [R,11], of every row: accumulate == 0 overwrites
 * the WHOLE row (zeros in the other columns), != 0 adds into those six columns and leaves the others alone. */
int evd_points(const float* ray_batch, int ncol, const float* z, long R, int S, float* pts, void* stream);
int evd_points_bwd(const float* z, const float* d_pts, long R, int S, int accumulate, float* d_ray_batch, void* stream);
/* z stratification, networks/renderer.py:163-178.  t_rand dev [R,S] is the explicit torch.rand draw
 * (required when cfg->perturb > 0).  -> z dev [R,S] */
int evd_sample_z(const evd_render_cfg* cfg, const float* ray_batch, int ncol, long R, const float* t_rand,
                 float* z, void* stream);

/* ---------------------------------------------------------------- NeRF backbone (networks/nerf.py) */
typedef struct evd_nerf evd_nerf;   /* opaque: packed MFMA-fragment weight streams on one device */

typedef struct {                    /* host float32 pointers, nn.Linear layout [out,in] (state_dict arrays) */
    int D, W, multires, multires_views, skip;
    int rgb_act, sigma_act;         /* EVD_ACT_* (nerf.py:34-35) */
    float rmnear;                   /* render_rmnearplane */
    const float *pts_w[EVD_MAX_LAYERS], *pts_b[EVD_MAX_LAYERS];
    const float *views_w, *views_b, *feature_w, *feature_b, *alpha_w, *alpha_b, *rgb_w, *rgb_b; /* rgb_b may be NULL */
    /* use_viewdirs=False networks (networks/nerf.py:41-44,158-160): views_w .. rgb_w NULL and output_linear given instead,
     * output_w [output_ch, W], output_b [output_ch], output_ch 4 or 5 (renderer.py:46: 5 with importance sampling; raw2outputs reads
     * channels 0..3 only).  Their ray batch has 8 columns (renderer.py:443-446).  Inference on the generic kernel (every mode but
     * EVD_PREC_F16C); no training entries.  In the reference this network only runs with extract_feature "before_linear" (kernel_use_awp):
     * nerf.py:159 asserts otherwise. */
    const float *output_w, *output_b;
    int output_ch;
} evd_nerf_desc;

int evd_nerf_create(const evd_nerf_desc* desc, evd_nerf** out);
void evd_nerf_destroy(evd_nerf* net);
/* Parameters live in one float32 arena, canonical order: pts_linears[l].weight, .bias for l < D, then views_linears.0,
 * feature_linear, alpha_linear, rgb_linear (weight, bias each; reference nn.Linear layouts, networks/nerf.py:14-44).
 * evd_nerf_param_blocks writes the arena offset of each of the 2 D + 8 tensors plus the total (up to `capacity` longs) and
 * returns 2 D + 8.  evd_nerf_load_params re-packs every weight stream of the network on the device from new values
 * (params: device float32 [evd_nerf_param_count]) -- what a training loop calls after optimizer.step() (run_nerf.py:601). */
long evd_nerf_param_count(const evd_nerf* net);
int evd_nerf_param_blocks(const evd_nerf* net, long* offsets, int capacity);
int evd_nerf_load_params(evd_nerf* net, const float* params, void* stream);
/* bytes of the packed weight stream for a precision (what one workgroup streams per sample tile) */
size_t evd_nerf_stream_bytes(const evd_nerf* net, int precision);

/* NeRF.mlpforward + NeRF.eval, networks/nerf.py:46-72,131-162, fused with the point computation
 * pts = o + d z (renderer.py:180) and both positional encodings.  ray_batch dev [R,11] ([R,8] for a use_viewdirs=False network), z dev [R,S]
 * -> raw dev [R,S,4] = (rgb, alpha);  feature dev [R,S,W] optional: feature_kind 1 = "after_linear"
 * (nerf.py:149-150), 2 = "before_linear" (:141-142). */
int evd_nerf_mlp(const evd_nerf* net, int precision, const float* ray_batch, const float* z, long R, int S,
                 float* raw, float* feature, int feature_kind, void* stream);

/* Training variant of evd_nerf_mlp (the forward half of the autograd graph of NeRF.mlpforward, networks/nerf.py:46-72, that
 * run_nerf.py:593-601 differentiates): same raw, and every layer's activations are kept in `store` (device,
 * evd_nerf_train_store_bytes(R * S) bytes, fragment layout of csrc/nerf_mlp.h) for evd_nerf_mlp_backward.
 * Built for precision EVD_PREC_F16 / EVD_PREC_BF16 and EVD_PREC_F16X3 on the netdepth 8, netwidth 256, skips [4] network;
 * EVD_E_INVALID otherwise.  EVD_PREC_F16X3 is the float32-grade training mode (the reference trains in float32, run_nerf.py:593-601):
 * every stored fragment is a (hi, lo) float16 pair, every product of the forward, the dgrad and the wgrad kernels is the 3-MFMA
 * split product -- twice the store (evd_nerf_train_store_bytes_prec), gradients equal to float32 autograd to ~1e-5.
 * evd_nerf_train_store_bytes = the half-precision size (kept for existing callers). */
size_t evd_nerf_train_store_bytes(long nsamp);
size_t evd_nerf_train_store_bytes_prec(int precision, long nsamp);
int evd_nerf_mlp_train(const evd_nerf* net, int precision, const float* ray_batch, const float* z, long R, int S,
                       float* raw, void* store, size_t store_bytes, void* stream);

/* Gradients of the network parameters, device float32 in the reference's nn.Linear layouts ([out, in] weights, [out] biases;
 * the same fields as evd_nerf_desc).  NULL = not wanted.  accumulate == 0: every wanted block is overwritten; != 0: the gradients are
 * ADDED into it (a training loop's persistent .grad buffers, several backward calls per optimizer step: run_nerf.py:593-601 runs one
 * loss.backward() over three renders). */
typedef struct {
    float *pts_w[EVD_MAX_LAYERS], *pts_b[EVD_MAX_LAYERS];
    float *views_w, *views_b, *feature_w, *feature_b, *alpha_w, *alpha_b, *rgb_w, *rgb_b;
    int accumulate;
} evd_nerf_grads;

/* Backward of evd_nerf_mlp_train: d_raw dev [R,S,4] (d loss / d raw) -> parameter gradients, what torch autograd computes for
 * NeRF.mlpforward (networks/nerf.py:46-72) under loss.backward() (run_nerf.py:593-601).  `store` is the one the forward
 * filled (it is consumed: the gradient fragments are written into it).  Arithmetic: MFMA operands in the forward's half
 * precision under a power-of-two loss scale chosen from max |d_raw|, float32 accumulation; the scale is removed before the
 * gradients are written.  d_pts / d_dirs dev [R*S,3] (NULL = not wanted): d loss / d sample position through PE(pts) (both
 * pts_linears[0] and the skip layer) and / d view direction through PE(dirs), per sample -- they need the forward's positions
 * pts dev [R*S,3] (= o + d z) and viewdirs (rows of vd_stride floats per ray).  workspace: evd_nerf_backward_workspace_bytes(). */
size_t evd_nerf_backward_workspace_bytes(void);
int evd_nerf_mlp_backward(const evd_nerf* net, int precision, const float* d_raw, long R, int S, void* store, size_t store_bytes,
                          const evd_nerf_grads* grads, const float* pts, const float* viewdirs, int vd_stride, float* d_pts,
                          float* d_dirs, void* workspace, size_t workspace_bytes, void* stream);

/* raw2outputs: networks/nerf.py:74-129 (sigma_ch 3, rgb_ch0 0) and networks/pdrf/voxnerf.py:153-201
 * (sigma_ch 0, rgb_ch0 1).  raw dev [R,S,C], z dev [R,S], rays_d dev rows of rays_d_stride floats.
 * noise dev [R,S-1] optional (explicit randn*raw_noise_std draw).  rmnear_thresh <= 0 disables the
 * eval near-plane mask.  Outputs: out_map [R,n_rgb], density [R,S-1], acc [R], weights [R,S],
 * depth [R], fmap [R,F] = sum_s w feature[R,S,F]. */
int evd_raw2outputs(const float* raw, const float* z, const float* rays_d, int rays_d_stride, long R, int S, int C,
                    int sigma_ch, int rgb_ch0, int n_rgb, int rgb_act, int sigma_act, int white_bkgd,
                    float rmnear_thresh, const float* noise,
                    float* out_map, float* density, float* acc, float* weights, float* depth,
                    const float* feature, int F, float* fmap, void* stream);

/* Backward of evd_raw2outputs (first slice of the training path; torch autograd provides this in the reference):
 * g_map dev [R,3], g_depth dev [R], g_acc dev [R], g_weights dev [R,S] = dL/d(out_map, depth, acc, weights), any may be
 * NULL (= zero) -> d_raw dev [R,S,4] = dL/d raw.  z and rays_d are sampling inputs and get no gradient (sample_pdf is
 * detached in the reference, renderer.py:201).  Built for C = 4, n_rgb = 3, S <= 256. */
int evd_raw2outputs_bwd(const float* raw, const float* z, const float* rays_d, int rays_d_stride, long R, int S, int C,
                        int sigma_ch, int rgb_ch0, int n_rgb, int rgb_act, int sigma_act, int white_bkgd, float rmnear_thresh,
                        const float* noise, const float* g_map, const float* g_depth, const float* g_acc, const float* g_weights,
                        float* d_raw, void* stream);
/* The same with the gradient w.r.t. the ray directions, d_rays_d dev [R, >= 3] (row stride d_rays_d_stride) or NULL: alpha depends on
 * rays_d through dists * |rays_d| (nerf.py:104, voxnerf.py:179), which the blur batch's learnable ray warp differentiates
 * (run_nerf.py:593): dL/d rays_d = (sum_i dL/d density_i * density_i) * rays_d / |rays_d|^2. */
int evd_raw2outputs_bwd_rays(const float* raw, const float* z, const float* rays_d, int rays_d_stride, long R, int S, int C,
                             int sigma_ch, int rgb_ch0, int n_rgb, int rgb_act, int sigma_act, int white_bkgd, float rmnear_thresh,
                             const float* noise, const float* g_map, const float* g_depth, const float* g_acc, const float* g_weights,
                             float* d_raw, float* d_rays_d, int d_rays_d_stride, void* stream);

/* sample_pdf, utils/rays.py:149-193, called as in renderer.py:200-201,230-231 with bins = z_mid and
 * weights[...,1:-1]: z dev [R,S], weights dev [R,S] -> z_samples dev [R,N].  det != 0 => u = linspace;
 * else u dev [R,N].  Also returns, fused: z_merged dev [R,S+N] = sort(cat(z, z_samples)) (renderer.py:205,234),
 * order dev int32 [R,S+N] (index into the concatenation, as torch.sort's second result), z_std dev [R]
 * (renderer.py:250). */
int evd_sample_pdf_merge(const float* z, const float* weights, long R, int S, int N, int det, const float* u,
                         float* z_samples, float* z_merged, int* order, float* z_std, void* stream);

typedef struct {        /* result dict of render_rays, networks/renderer.py:242-257; dev pointers or NULL */
    float *rgb, *depth, *acc;                   /* rgb_map [R,3], depth_map [R], acc_map [R] */
    float *z_vals, *weights;                    /* [R, N_samples+N_importance] */
    float *rgb0, *depth0, *acc0, *z_std;        /* coarse pass (N_importance > 0) */
    float *z_vals0, *weights0;                  /* [R, N_samples] */
    float *feature;                             /* per-sample feature of the last pass [R,S_final,F] */
    float *raw;                                 /* raw network output of the last pass [R,S_final,4] */
    int feature_kind;                           /* NeRF: 1 after_linear, 2 before_linear */
} evd_render_out;

size_t evd_nerf_render_workspace_bytes(const evd_render_cfg* cfg, long R);
/* NeRFAll.render_rays for mode='nerf' (networks/renderer.py:129-264, else-branch :218-240): ray_batch dev [R,11].
 * fine may be NULL iff N_importance == 0.  Explicit random draws (the reference calls torch.rand / randn
 * inside): t_rand dev [R,N_samples] and u dev [R,N_importance] when perturb > 0; noise0 dev [R,N_samples-1] /
 * noise1 dev [R,N_samples+N_importance-1] = randn * raw_noise_std (NULL = no density noise). */
int evd_nerf_render_rays(const evd_nerf* coarse, const evd_nerf* fine, const evd_render_cfg* cfg, const float* ray_batch,
                         long R, const float* t_rand, const float* u, const float* noise0, const float* noise1,
                         evd_render_out* out, void* workspace, size_t workspace_bytes, void* stream);
/* NeRFAll.render (networks/renderer.py:399-466) = evd_ray_batch + evd_nerf_render_rays: rays dev [R,3,2] */
int evd_nerf_render(const evd_nerf* coarse, const evd_nerf* fine, const evd_render_cfg* cfg, const float* rays, long R,
                    const float* t_rand, const float* u, const float* noise0, const float* noise1,
                    evd_render_out* out, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------- PDRF backbone (networks/pdrf/voxnerf.py) */
typedef struct evd_voxel evd_voxel; /* opaque: channel-last tri-plane grids + packed MLPs on one device */

typedef struct {                    /* host float32 pointers */
    int num_layers, hidden_dim, geo_feat_dim, num_layers_color, input_ch, multires, multires_views;
    int app_dim, n_comp[3], grid[3], app_act, rgb_act, sigma_act, composite_feature;
    float aabb[6], rmnear;
    const float *sigma_w[EVD_MAX_LAYERS];
    const float *color_w[EVD_MAX_LAYERS], *color_b[EVD_MAX_LAYERS];
    const float *plane[3], *line[3], *basis;
} evd_voxel_desc;

int evd_voxel_create(const evd_voxel_desc* desc, evd_voxel** out);
void evd_voxel_destroy(evd_voxel* v);
/* VoxelNeRFBase.sample / compute_appfeature, voxnerf.py:203-208,132-151.  pts dev [n,3] -> out dev [n,out_stride]
 * written at column out_col (so coarse and fine features can share one [n,64] buffer, renderer.py:195) */
int evd_voxel_sample(const evd_voxel* v, const float* pts, long n, float* out, int out_stride, int out_col, void* stream);
/* The same as the c2f renderer runs it in arithmetic mode `precision`: EVD_PREC_F16 / EVD_PREC_BF16 gather the float16 copies of the
 * grids (half the bytes; the level networks round their inputs to 2^-11 / 2^-8 anyway), the float32-grade modes the float32 grids.
 * What the training forward calls, so that it is the inference render's arithmetic. */
int evd_voxel_sample_prec(const evd_voxel* v, int precision, const float* pts, long n, float* out, int out_stride, int out_col, void* stream);
/* VoxelNeRFBase.forward, voxnerf.py:210-259.  pts dev [R,S,3], viewdirs dev rows of vd_stride floats,
 * fts dev [R,S,F] -> color [R,3], depth [R], acc [R], weights [R,S], feature [R,S,geo].
 * A level created with composite_feature != 0 (kernel_type PBE, voxnerf.py:223-239) composites its geo features first and runs the
 * colour network per ray: `feature` is then the COMPOSITED map [R,geo].  Inference only; the workspace query includes its scratch. */
int evd_voxel_forward(const evd_voxel* v, int precision, const float* pts, const float* viewdirs, int vd_stride, const float* fts, int F,
                      const float* z, const float* rays_d, int rays_d_stride, long R, int S, int is_train,
                      float* color, float* depth, float* acc, float* weights, float* feature,
                      void* workspace, size_t workspace_bytes, void* stream);
size_t evd_voxel_forward_workspace_bytes(const evd_voxel* v, long R, int S);
size_t evd_c2f_render_workspace_bytes(const evd_voxel* coarse, const evd_voxel* fine, const evd_render_cfg* cfg, long R);
/* NeRFAll.render_rays for mode='c2f' (networks/renderer.py:182-217); arguments as evd_nerf_render_rays.
 * out->feature receives the fine level's per-sample geo features [R,S_final,geo_feat_dim] (what AWP consumes). */
int evd_c2f_render_rays(const evd_voxel* coarse, const evd_voxel* fine, const evd_render_cfg* cfg, const float* ray_batch, long R,
                        const float* t_rand, const float* u, const float* noise0, const float* noise1,
                        evd_render_out* out, void* workspace, size_t workspace_bytes, void* stream);
/* NeRFAll.render for mode='c2f' = evd_ray_batch + evd_c2f_render_rays */
int evd_c2f_render(const evd_voxel* coarse, const evd_voxel* fine, const evd_render_cfg* cfg, const float* rays, long R,
                   const float* t_rand, const float* u, const float* noise0, const float* noise1,
                   evd_render_out* out, void* workspace, size_t workspace_bytes, void* stream);
/* The coarse feature rows of the merged sample set, renderer.py:209-213: torch.cat([ft_comb0, ft_comb1], 1) gathered by the sort order
 * (`order` of evd_sample_pdf_merge): old dev [R,S,F], fresh dev [R,N,F], order dev int32 [R,S+N] -> out dev rows of out_stride floats
 * (columns 0..F-1 written).  evd_c2f_render_rays does this internally; the entry (and its backward: a permutation, d_out rows of
 * d_stride floats -> d_old dev [R,S,F], d_fresh dev [R,N,F], every row written once) serves the training path under autograd. */
int evd_merge_features(const float* old, const float* fresh, const int* order, long R, int S, int N, int F, float* out, int out_stride, void* stream);
int evd_merge_features_bwd(const float* d_out, int d_stride, const int* order, long R, int S, int N, int F, float* d_old, float* d_fresh, void* stream);
/* TV_loss_app, voxnerf.py:126-130 (TVLoss :306-324): sum over planes*1e-2 + lines*1e-3 -> out dev [1] (float) */
int evd_voxel_tv_loss(const evd_voxel* v, float* out, void* stream);

/* ---- training one PDRF level's sigma / colour networks -------------------------------------------------
 * Parameters live in one float32 arena, canonical order sigma_net.0.weight, sigma_net.1.weight, color_net.{0,1,2}.{weight,bias}
 * (reference nn.Linear layouts, voxnerf.py:60-84); evd_voxel_param_blocks writes the 8 offsets + the total and returns 8;
 * evd_voxel_load_params re-packs every weight stream of the level on the device (after optimizer.step()).
 * evd_voxel_mlp_train = the per-sample part of VoxelNeRFBase.forward (voxnerf.py:210-221,240-254): raw dev [R,S,4] =
 * (sigma, sigmoid(colour)), every layer's activations kept in `store` (evd_voxel_train_store_bytes); feature dev [R,S,geo]
 * (NULL = not wanted; fine level) = the per-sample geo features AWP consumes (voxnerf.py:221), whose gradient
 * evd_voxel_mlp_backward takes back as d_feature (NULL = none).
 * evd_voxel_mlp_backward: d_raw, raw dev [R,S,4] -> parameter gradients (overwritten) and d_fts dev [R*S, d_fts_stride]
 * (columns 0 .. ft_dim-1 overwritten; NULL = not wanted), the gradient of the sampled features that evd_voxel_sample_bwd
 * scatters into the grids.  d_pts / d_dirs dev [R*S, 3] (NULL = not wanted; need the forward's pts / viewdirs): the gradient
 * that reaches the sample position through PE(pts) and the view direction through PE(dirs) (per sample: the caller sums over a
 * ray's samples) -- what carries the loss back to the rays, i.e. to the blur kernel's camera motion.  awp_store (NULL = none; fine
 * level): the store of the fused AWP embedding AFTER evd_awp_embed_backward ran on the same samples -- its d geo fragments are added
 * to the geo features' gradient without ever becoming a float32 [R*S,128] tensor (below).
 * Built for EVD_PREC_F16 / EVD_PREC_BF16 and the float32-grade EVD_PREC_F16X3 (as evd_nerf_mlp_train; awp_store is a half-precision
 * coupling: in f16x3 the AWP consumer's gradient comes back as d_feature rows), both shipped levels (64/15/32 and 256/128/64). */
typedef struct { float *sigma_w[2], *color_w[3], *color_b[3]; int accumulate; /* as in evd_nerf_grads */ } evd_voxel_grads;
int evd_voxel_geo_feat_dim(const evd_voxel* v);
long evd_voxel_param_count(const evd_voxel* v);
int evd_voxel_param_blocks(const evd_voxel* v, long* offsets, int capacity);
int evd_voxel_load_params(evd_voxel* v, const float* params, void* stream);
size_t evd_voxel_train_store_bytes(const evd_voxel* v, long nsamp);         /* the f16 / bf16 size */
size_t evd_voxel_train_store_bytes_prec(const evd_voxel* v, int precision, long nsamp);   /* EVD_PREC_F16X3: (hi, lo) fragments, twice that */
size_t evd_voxel_backward_workspace_bytes(void);
int evd_voxel_mlp_train(const evd_voxel* v, int precision, const float* pts, const float* viewdirs, int vd_stride, const float* fts,
                        int ft_stride, long R, int S, float* raw, float* feature, void* store, size_t store_bytes, void* stream);
int evd_voxel_mlp_backward(const evd_voxel* v, int precision, const float* d_raw, const float* raw, const float* d_feature,
                           const void* awp_store, size_t awp_store_bytes, long R, int S, void* store, size_t store_bytes, const evd_voxel_grads* grads, float* d_fts, int d_fts_stride, const float* pts,
                           const float* viewdirs, int vd_stride, float* d_pts, float* d_dirs, void* workspace,
                           size_t workspace_bytes, void* stream);

/* ---- training the PDRF grids (scatter-add into the tri-planes) -----------------------------------------
 * The library keeps planes / lines CHANNEL-LAST: plane[i] [grid[m1]][grid[m0]][C_i] (reference app_plane.i is [1,C,H,W],
 * voxnerf.py:112-113), line[i] [grid[vec]][C_i], basis [app_dim][sum C] (= basis_mat.weight).  Parameters and gradients of
 * the training path use THAT layout (a permute away from the state dict).
 * evd_voxel_grid_sizes: element counts, sizes[0..2] planes, [3..5] lines, [6] basis.
 * evd_voxel_get_grids / evd_voxel_load_grids: copy the grids out to / in from caller-owned device buffers (load also
 * refreshes the float16 copies the half-precision modes read) -- what a training loop calls after optimizer.step(). */
typedef struct { float *plane[3], *line[3], *basis; } evd_voxel_grid_grads;
int evd_voxel_grid_sizes(const evd_voxel* v, long* sizes);
int evd_voxel_get_grids(const evd_voxel* v, float* const* plane, float* const* line, float* basis, void* stream);
int evd_voxel_load_grids(evd_voxel* v, const float* const* plane, const float* const* line, const float* basis, void* stream);
/* Backward of evd_voxel_sample (VoxelNeRFBase.sample / compute_appfeature, voxnerf.py:132-151,203-208; app_actfn none):
 * d_out dev rows of d_stride floats, the app_dim gradient columns start at d_col -> gradients ADDED into g (caller zeroes;
 * NULL = not wanted) with float32 hardware atomics; like the reference's grid_sample backward (voxnerf.py:144) the summation
 * order is not deterministic.  d_pts dev [n,3] (NULL = not wanted) is OVERWRITTEN with d loss / d pts through the bilinear /
 * linear interpolation weights (ATen grid_sample backward semantics: taps outside the grid contribute nothing). */
int evd_voxel_sample_bwd(const evd_voxel* v, const float* pts, long n, const float* d_out, int d_stride, int d_col,
                         const evd_voxel_grid_grads* g, float* d_pts, void* stream);
/* The same with caller scratch (evd_voxel_sample_bwd_workspace_bytes(v, n), ~430 B per sample; n_comp in {16, 32, 64}, else the
 * call is evd_voxel_sample_bwd): the HYBRID form -- the plane taps stay direct float atomics, the line taps (a third of the atomic
 * requests, onto a few hundred cells) are summed per 2048 samples in LDS slices of the line gradients (64-bit fixed point, exact
 * conversion of every float32 contribution above 2^-49 of the chunk's maximum) and added once: 24-27 % faster, same sums to float32
 * rounding of the summation order.  With workspace NULL / 0 bytes this IS evd_voxel_sample_bwd.  (Developer switch EVD_SCATTER=binned:
 * the experimental sort + LDS-tile form, csrc/kernel_voxel_scatter.hip.) */
size_t evd_voxel_sample_bwd_workspace_bytes(const evd_voxel* v, long n);
int evd_voxel_sample_bwd_ws(const evd_voxel* v, const float* pts, long n, const float* d_out, int d_stride, int d_col,
                            const evd_voxel_grid_grads* g, float* d_pts, void* workspace, size_t workspace_bytes, void* stream);
/* The same in the arithmetic mode `precision` of the forward gather (evd_voxel_sample_prec): where that forward interpolated the FLOAT16
 * copies of the grids (f16 / bf16 on every level, f16c on the fine level of a c2f pair), the backward's re-gather of the plane / line values
 * (d plane = d coef x line value, d line = d coef x plane value, d pts) reads the same copies -- the gradient of the function the forward
 * computed, at half the gather's loads and bytes; the other modes are evd_voxel_sample_bwd_ws. */
int evd_voxel_sample_bwd_prec(const evd_voxel* v, int precision, const float* pts, long n, const float* d_out, int d_stride, int d_col,
                              const evd_voxel_grid_grads* g, float* d_pts, void* workspace, size_t workspace_bytes, void* stream);
/* d_loss[0] * d TV_loss_app / d grid added into g (voxnerf.py:126-130, 306-324); d_loss is a DEVICE scalar (no host sync) */
int evd_voxel_tv_loss_bwd(const evd_voxel* v, const float* d_loss, const evd_voxel_grid_grads* g, void* stream);

/* ---------------------------------------------------------------- loss-side pixel ops */
/* RigidBlurringModel.rbk_weighted_sum, networks/dpnerf/blurmodel.py:112-127 (and renderer.py:299,332-354):
 * out[r,c] = sum_p ccw[r,p] x[r*P+p, c] */
int evd_weighted_sum(const float* x, const float* ccw, long R, int P, int C, float* out, void* stream);

typedef struct evd_crf evd_crf;     /* opaque: CRF MLP parameters (networks/tonemapping.py:7-93) */
typedef struct {
    int map_type;                   /* 0 none, 1 gamma, 2 learn */
    float gamma;
    int extra_features;
    const float *w[4], *b[4];       /* host: linear.{0,2,4,6}.{weight,bias} when map_type == 2 */
} evd_crf_desc;
int evd_crf_create(const evd_crf_desc* desc, evd_crf** out);
void evd_crf_destroy(evd_crf* crf);
/* CRF.forward (+ rec601/rec709/avg luma of TonemappingTransform.encode_luma, tonemapping.py:120-139).
 * x dev [n,3]; feat dev NULL | [n,E] (feat_per_channel 0) | [n,3,E] (1).  luma < 0: out [n,3];
 * luma 0/1/2 = rec601/rec709/avg: out [n,1]. */
int evd_crf_forward(const evd_crf* crf, const float* x, const float* feat, int feat_per_channel, int skip_learn,
                    int luma, long n, float* out, void* stream);

/* Fused blur-loss reduction for one pixel batch (spec: run_nerf.py:443-497 + rbk_weighted_sum):
 *   rgb   = sum_p w1[r,p] rgb_p[r,p]       rgb1 = sum_p w1[r,p] rgb0_p[r,p]     awp = sum_p w2[r,p] rgb_p[r,p]
 *   partial[0..2] = sum_r |crf(rgb)-tgt|^2, |crf(rgb1)-tgt|^2, |crf(awp)-tgt|^2   (3 channels summed)
 *   partial[3..4] = sum_r |crf(rgb_p[r,0])-tgt0|^2, |crf(rgb0_p[r,0])-tgt0|^2     (pts0 / EDI prior terms)
 *   partial[5]    = 3 R  (element count)
 * rgb0_p, w2, tgt0 may be NULL (their partials stay 0).  Accumulates (atomicAdd) into partial dev [8]:
 * zero it once per step; ranks all-reduce the 8 floats (evdeblurnerf_amd/dist.py).  Optionally writes
 * the blended colours rgb/rgb1/awp dev [R,3]. */
int evd_blur_loss_reduce(const evd_crf* crf_rgb, int skip_learn, const float* rgb_p, const float* rgb0_p,
                         const float* w1, const float* w2, const float* tgt, const float* tgt0, long R, int P,
                         float* partial, float* out_rgb, float* out_rgb1, float* out_awp, void* stream);

/* Backward of evd_blur_loss_reduce (identity / gamma response on the image branch, as in every shipped config):
 * g_partial host[5] = dL/d partial[0..4] -> d_rgb_p, d_rgb0_p dev [R,P,3] (d_rgb0_p may be NULL), d_w1, d_w2 dev [R,P] or NULL. */
int evd_blur_loss_bwd(const evd_crf* crf_rgb, int skip_learn, const float* rgb_p, const float* rgb0_p, const float* w1, const float* w2,
                      const float* tgt, const float* tgt0, long R, int P, const float* g_partial, float* d_rgb_p, float* d_rgb0_p,
                      float* d_w1, float* d_w2, void* stream);
/* the same with dL/d partial read from DEVICE memory (g_partial_dev dev[>=5]): what an autograd node calls, no host copy in the backward */
int evd_blur_loss_bwd_dev(const evd_crf* crf_rgb, int skip_learn, const float* rgb_p, const float* rgb0_p, const float* w1, const float* w2,
                          const float* tgt, const float* tgt0, long R, int P, const float* g_partial_dev, float* d_rgb_p, float* d_rgb0_p,
                          float* d_w1, float* d_w2, void* stream);

/* Fused event-loss reduction (spec: run_nerf.py:518-570, utils/events.py:260-284):
 *   bii = thr_neg*cum_neg + thr_pos*cum_pos;  feat = (cum_neg, cum_pos) per event ("pos-neg") or scattered to the
 *   event's colour channel ("color-pos-neg", color_mask != NULL); luma = CRF_event(rgb) -> rec601 luma, or the
 *   masked channel when tonemap_only;  pred = log(l_end+1e-5) - log(l_start+1e-5);
 *   partial[0] += sum w (pred-bii)^2 (fine), partial[1] += same for the coarse pair, partial[2] += sum w.
 * start/end/start0/end0 dev [N,3] linear rgb; cum_neg/cum_pos dev [N]; color_mask dev uint8 [N,3] or NULL;
 * color_weight host[3] or NULL.  start0/end0 may be NULL. */
int evd_event_loss_reduce(const evd_crf* crf_ev, int skip_learn, int add_bii_feat, int tonemap_only,
                          const float* start, const float* end, const float* start0, const float* end0,
                          const float* cum_neg, const float* cum_pos, float thr_neg, float thr_pos,
                          const unsigned char* color_mask, const float* color_weight, long N,
                          float* partial, void* stream);

/* Backward of evd_event_loss_reduce: gradients of  g_fine * partial[0] + g_coarse * partial[1]  w.r.t. the colour inputs
 * (d_start, d_end, d_start0, d_end0 dev [N,3]; the *0 pair may be NULL) and w.r.t. the learnable event-CRF parameters
 * (d_params dev [evd_crf_param_count()] or NULL; layout: linear.0.weight as [16][8] rows zero-padded from 1 + extra_features,
 * linear.0.bias [16], linear.2.weight [16][16], linear.2.bias, linear.4.weight [16][16], linear.4.bias, linear.6.weight [16],
 * linear.6.bias [1]).  d_params is overwritten. */
int evd_crf_param_count(void);
/* a learn CRF's parameters to / from a HOST array of evd_crf_param_count() floats in that layout (the handle keeps them on the host:
 * they travel as a kernel argument); load is what a loop that trains the event-CRF calls after optimizer.step() */
int evd_crf_get_params(const evd_crf* crf, float* host_params);
int evd_crf_load_params(evd_crf* crf, const float* host_params);
int evd_event_loss_bwd(const evd_crf* crf_ev, int skip_learn, int add_bii_feat, int tonemap_only,
                       const float* start, const float* end, const float* start0, const float* end0,
                       const float* cum_neg, const float* cum_pos, float thr_neg, float thr_pos,
                       const unsigned char* color_mask, const float* color_weight, long N, float g_fine, float g_coarse,
                       float* d_start, float* d_end, float* d_start0, float* d_end0, float* d_params, void* stream);
/* the same with (g_fine, g_coarse) = g_partial_dev[0], [1] read from DEVICE memory */
int evd_event_loss_bwd_dev(const evd_crf* crf_ev, int skip_learn, int add_bii_feat, int tonemap_only,
                           const float* start, const float* end, const float* start0, const float* end0,
                           const float* cum_neg, const float* cum_pos, float thr_neg, float thr_pos,
                           const unsigned char* color_mask, const float* color_weight, long N, const float* g_partial_dev,
                           float* d_start, float* d_end, float* d_start0, float* d_end0, float* d_params, void* stream);

/* AdaptiveWeightProposal.feature_integration, networks/dpnerf/awp.py:49-77 (the compositing scan of the AWP consumer of
 * the path's per-sample features): feat dev [N,S,C] (N = rays x sub-exposures; every channel is its own density),
 * z dev [N,S], rays_d dev [N,3] -> out dev [N,C].  Restated as written: the last sample gets alpha 0 (awp.py:67) and the
 * cumprod of awp.py:69-73 runs along the channel axis of the previous sample's row. */
int evd_awp_feature_integration(const float* feat, const float* z, const float* rays_d, long N, int S, int C, float* out, void* stream);
/* Its backward (what torch.autograd does behind awp.py:98-104 during training, run_nerf.py:593-601): d out dev [N,C] -> d feat dev
 * [N,S,C]; d z dev [N,S] and d rays_d dev [N,3] where wanted (null: skipped; the distances are (z[s+1] - z[s]) |rays_d|, awp.py:61-63). */
int evd_awp_feature_integration_bwd(const float* feat, const float* z, const float* rays_d, const float* d_out, long N, int S, int C,
                                    float* d_feat, float* d_z, float* d_rays_d, void* stream);

/* MotionAggregationModule / CorrelationModule, the per-sample part (networks/dpnerf/mam.py:72-74: self.linear on x_local = h_local
 * [R P, S, 64]; :29-33: line_conv_att, softmax over the samples (dim -1) and over the sub-exposures (dim -2), the two weighted sums).
 * The linear map commutes with the softmax-weighted sums and a constant leaves a softmax unchanged, so with u = W^T v dev [64]
 * (W = MAM.linear.weight [32,64], v = Corr.line_conv_att.weight [32]):  logit = u . h_local;  alpha = softmax_s, beta = softmax_p;
 * h_inter dev [R,P,64] = sum_s alpha h_local, h_intra dev [R,S,64] = sum_p beta h_local -- the caller applies W, b to those
 * (curver_inter = W h_inter + b, curves_intra = W h_intra + b, mam.py:31-32).  alpha, beta dev [R,P,S] are kept for the backward.
 * C must be 64 (the embedding's width), P <= 16, S <= 512. */
int evd_mam_local_forward(const float* h_local, const float* u, long R, int P, int S, int C, float* h_inter, float* h_intra, float* alpha,
                          float* beta, void* stream);
/* Its backward (torch.autograd behind mam.py:29-33,72-74 in training, run_nerf.py:593-601): d h_inter dev [R,P,64], d h_intra dev
 * [R,S,64] -> d h_local dev [R P, S, 64] (accumulate = 0: written; != 0: ADDED to what is there -- h_local has a second consumer, the
 * feature integration, whose backward writes the buffer first) and d u as per-ray partials dev [R,64] (the caller sums them).
 * d_h_absmax (optional): dev word, set to 0 by this entry (on the stream) and raised to the float bits of max |d h_local| as written here (the loss scale the
 * embedding's backward needs: saves it a pass over the tensor). */
int evd_mam_local_backward(const float* h_local, const float* u, const float* alpha, const float* beta, const float* h_inter,
                           const float* h_intra, const float* d_inter, const float* d_intra, long R, int P, int S, int C, float* d_h_local,
                           float* d_u_partial, int accumulate, unsigned* d_h_absmax, void* stream);

/* The PER-RAY remainder of the adaptive weight proposal as kernels: networks/dpnerf/awp.py:89-95 (direction encoding of the first
 * sub-exposure's normalised ray direction, concatenated behind view_feature), :104-109 (motion_feature_embed_layer: n_mot x Linear + ReLU on
 * [integrated features | view_embedded]), networks/dpnerf/mam.py:35-53 (CorrelationModule.forward behind its per-sample part: conva /
 * convb / convc, the two softmax(bmm) attention maps over the P sub-exposures and the S samples, convn / convl, convd = Conv1d +
 * BatchNorm1d, residual, leaky_relu 0.2) and awp.py:112-115 (mean over P, w_linear, sigmoid, normalisation) -- in the reference ~100 small
 * launches each way.  Two launches forward (the BatchNorm statistics are over ALL rays: per-workgroup partial sums in the first, folded by
 * the second), three backward.  Inputs per ray: h dev [R,P,64] (evd_awp_feature_integration's output), view_feature dev [R,VF] (NULL when
 * VF = 0), rays_d dev [R P,3], h_inter dev [R,P,64] / h_intra dev [R,S,64] (evd_mam_local_forward's outputs; MAM.linear is applied here).
 * Built for W_sam 64, W_mot 32 (kernel_awp_sam_emb_width / kernel_awp_mot_emb_width of the shipped configs), n_mot <= 4, P <= 16,
 * dir_freqs 0..4 (get_embedder(ray_dir_freq): [d, sin(2^k d), cos(2^k d)...], 3 + 6 dir_freqs columns; -1: no direction columns), VF <= 64,
 * and S as far as the per-ray working set fits the 160 KiB LDS (S <= 128 for the backward at P = 10); EVD_E_INVALID otherwise.
 *
 * params: HOST array of evd_awp_tail_num_params(n_mot) DEVICE pointers, float32, nn.Module layouts ([out, in]):
 *   motion_feature_embed_layer.{l}.weight, .bias (l = 0..n_mot-1; layer 0 is [32, 64 + VF + 3 + 6 dir_freqs]),  MAM.linear.weight [32,64],
 *   .bias,  Corr.conva / convb / convc .weight [16,32],  Corr.convn / convl .weight [16,16],  Corr.convd.0.weight [32,32],
 *   Corr.convd.1.weight, .bias [32],  w_linear.weight [P,32], .bias [P].
 * d_params (backward): ONE flat dev float32 buffer of evd_awp_tail_param_count elements, the same tensors in the same order.
 * BatchNorm: training != 0 normalises with the batch statistics (biased variance) and, when bn_running_mean / _var are given, blends the
 * batch mean / UNBIASED variance into them with bn_momentum and adds 1 to *bn_num_batches (int64, may be NULL); training == 0 uses the
 * running estimates.  saved_y, saved_xg dev [R,P,32] and saved_stats dev [64] (mean, 1/sqrt(var + eps)) are kept for the backward;
 * saved_rays dev [R, evd_awp_tail_saved_floats(d)] (optional in both calls) takes every ray's forward state -- 55 KB per ray at P = 10,
 * S = 128 -- so that the backward does not run the forward a second time (measured: a third of its time). */
typedef struct {
    int P, S, VF, dir_freqs, n_mot, training;
    float bn_eps, bn_momentum;
} evd_awp_tail_desc;
int evd_awp_tail_num_params(int n_mot);
long evd_awp_tail_param_count(const evd_awp_tail_desc* d);
size_t evd_awp_tail_workspace_bytes(const evd_awp_tail_desc* d, long R, int backward);
long evd_awp_tail_saved_floats(const evd_awp_tail_desc* d);
int evd_awp_tail_forward(const evd_awp_tail_desc* d, const float* const* params, const float* h, const float* view_feature,
                         const float* rays_d, const float* h_inter, const float* h_intra, long R, float* bn_running_mean,
                         float* bn_running_var, long long* bn_num_batches, float* out, float* saved_y, float* saved_xg, float* saved_stats,
                         float* saved_rays, void* workspace, size_t workspace_bytes, void* stream);
/* d_out dev [R,P] -> d_h dev [R,P,64], d_view_feature dev [R,VF] (NULL when VF = 0), d_rays_d dev [R P,3] (rows other than a ray's first
 * are written as zeros), d_h_inter dev [R,P,64], d_h_intra dev [R,S,64], d_params (above); all written, not accumulated. */
int evd_awp_tail_backward(const evd_awp_tail_desc* d, const float* const* params, const float* h, const float* view_feature,
                          const float* rays_d, const float* h_inter, const float* h_intra, long R, const float* saved_y,
                          const float* saved_xg, const float* saved_stats, const float* saved_rays, const float* d_out, float* d_h,
                          float* d_view_feature,
                          float* d_rays_d, float* d_h_inter, float* d_h_intra, float* d_params, void* workspace, size_t workspace_bytes,
                          void* stream);

/* The backward of h_local's two consumers in one launch -- evd_awp_feature_integration_bwd (awp.py:49-77; feat = h_local, N = R P rows,
 * d_integrated dev [R P, 64] = d out) and evd_mam_local_backward (above) --: h_local is read once and d h_local dev [R P, S, 64] written
 * once with the SUM of both shares; d_z dev [R P, S] and d_rays_d dev [R P, 3] (the integration's; null = not wanted), d_u_partial dev
 * [R, 64], d_h_absmax as in evd_mam_local_backward.  C must be 64, P <= 16, S <= 512. */
int evd_awp_local_consumers_backward(const float* h_local, const float* u, const float* alpha, const float* beta, const float* h_inter,
                                     const float* h_intra, const float* d_inter, const float* d_intra, const float* z, const float* rays_d,
                                     const float* d_integrated, long R, int P, int S, int C, float* d_h_local, float* d_u_partial, float* d_z,
                                     float* d_rays_d, unsigned* d_h_absmax, void* stream);

/* AdaptiveWeightProposal.sample_feature_embed_layer (networks/dpnerf/awp.py:36-37: D_sam x nn.Linear; :98-100: each followed by ReLU)
 * fused into one MFMA kernel that reads the fine level's geo features WHERE THEY ALREADY ARE: the reference writes them as
 * depth_feature [R P, S, 128] float32 (renderer.py:253-256) and runs four torch Linear + ReLU passes over it (:314); here the
 * training forward of the fine level keeps the geo features as MFMA fragments for its own backward (evd_voxel_mlp_train's store),
 * evd_awp_embed_forward reads those fragments and writes only h_local dev [n, W_sam] float32 -- the tensor awp.py:102
 * (feature_integration) and :113 (the MAM) consume.  Built for input_ch 128, W_sam 64, D_sam 4 (fine_geo_feat_dim,
 * kernel_awp_sam_emb_width / _depth of the shipped configs; EVD_E_INVALID otherwise), EVD_PREC_F16 / EVD_PREC_BF16.
 *   weights[l] host [W_sam, in_l] / biases[l] host [W_sam]: sample_feature_embed_layer.l.{weight,bias}
 *   evd_awp_embed_load_params: flat dev float32 arena W0, b0, W1, b1, ... (evd_awp_embed_param_count elements) -> every stream
 *   evd_awp_embed_forward: geo features EITHER as geo_rows dev [n,128] float32 OR as (fine, fine_store) = the fine level and the
 *     store its evd_voxel_mlp_train filled for the same n = R*S samples.  store (NULL: inference) keeps the activations for
 *     evd_awp_embed_backward (evd_awp_embed_store_bytes).
 *   evd_awp_embed_backward: d_h_local dev [n, W_sam] (d_h_absmax: dev word holding the float bits of max |d_h_local| when the producer
 *     took it -- evd_mam_local_backward does --, NULL = a pass over d_h_local finds it) -> parameter gradients (overwritten; NULL = not wanted) and the geo features'
 *     gradient, left as fragments in `store` for evd_voxel_mlp_backward(awp_store = store) and, where wanted, written as
 *     d_geo_rows dev [n,128] float32 (NULL = not wanted). */
typedef struct evd_awp_embed evd_awp_embed;
typedef struct { float *w[4], *b[4]; } evd_awp_embed_grads;
int evd_awp_embed_create(const float* const* weights, const float* const* biases, int input_ch, int width, int depth, evd_awp_embed** out);
void evd_awp_embed_destroy(evd_awp_embed* a);
long evd_awp_embed_param_count(const evd_awp_embed* a);
int evd_awp_embed_load_params(evd_awp_embed* a, const float* params, void* stream);
size_t evd_awp_embed_store_bytes(const evd_awp_embed* a, long nsamp);
size_t evd_awp_embed_backward_workspace_bytes(void);
int evd_awp_embed_forward(const evd_awp_embed* a, int precision, const float* geo_rows, const evd_voxel* fine, const void* fine_store,
                          size_t fine_store_bytes, long nsamp, float* h_local, void* store, size_t store_bytes, void* stream);
int evd_awp_embed_backward(const evd_awp_embed* a, int precision, const float* d_h_local, long nsamp, void* store, size_t store_bytes,
                           const evd_awp_embed_grads* grads, float* d_geo_rows, const unsigned* d_h_absmax, void* workspace,
                           size_t workspace_bytes, void* stream);

/* EDI prior (utils/edi.py:73-95): bii dev [steps-1, npix], blurry dev [npix] -> sharp dev [npix] */
int evd_edi_deblur(const float* blurry, const float* bii, int steps, long npix, float* sharp, void* stream);
/* brightness_increment_image with bilinear sub-pixel splat (utils/edi.py:7-70, grey events):
 * x,y dev [n] float, p dev int8 [n] -> image dev [h,w] (overwritten) */
int evd_edi_bii_image(const float* x, const float* y, const signed char* p, long n, int w, int h,
                      float c_pos, float c_neg, float* image, void* stream);

/* ---------------------------------------------------------------- event preprocessing
 * compute_successor, utils/events.py:72-120 (flat_xy = True): pixel_ids dev int32 [N] (y * w + x of every event, in stream
 * order), HW = number of pixels -> successor dev int64 [N] (index of the next event at the same pixel, or the event itself),
 * num_successors dev int32 [N], latest_seen dev int64 [HW] (the pixel's first event, -1 if none), first_seen dev int64 [HW]
 * (its last event).  Bit-identical to the reference loop. */
size_t evd_compute_successor_workspace_bytes(long N);
int evd_compute_successor(const int* pixel_ids, long N, long HW, long long* successor, int* num_successors,
                          long long* latest_seen, long long* first_seen, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------- event batch assembly (second half)
 * EventsDataset.sample_events, data/loader_events.py:259-304: for a batch of event ids the start / end event pair, the polarity sums
 * between them and the two rays of each pair -- the reference gathers on the device, round-trips the timestamps through
 * .cpu().numpy() for the pose interpolation (:280-283) and calls get_rays_pix twice (:292-295, utils/rays.py:25-36); here ONE launch on
 * resident tables.
 *   events        dev float64 [N, ncol], the reference's augmented table (:247): column 0 = coordinate id, ncol-3 = timestamp,
 *                 ncol-2 = polarity, ncol-1 = successor index (compute_successor)
 *   id_to_coords  dev float32 [Ncoords, 2] (x, y)             id_to_color_map  dev uint8 [Ncoords, 3] or NULL (colour events)
 *   poses         dev float32 [N, 3, 4]: c2w of EVERY event at its own timestamp (the reference's interpolate_poses -- scipy Slerp +
 *                 cubic spline on the CPU, out of scope -- evaluated once per dataset instead of once per batch; 48 bytes per event)
 *   events_ids    dev int64 [n]          hops  dev int64 [n] or NULL
 *   hops == NULL: the branch of every shipped config (event_accumulate_step_range [0, 0], :272-276): end = the successor,
 *                 pos / neg = its polarity where positive / where not.
 *   hops != NULL: gather_successor (utils/events.py:221-257): hops[i] + 1 successor steps, the polarities of the visited events summed
 *                 by sign; a step that leaves [0, N) gives successor -1 and zero sums (:252-255).  The random hop counts of :265-268 are
 *                 the caller's draw.
 *   add_halfpix   the reference passes integer_coords (:292)
 * outputs (dev): rays_start / rays_end float32 [n, 3, 2] (origin, direction in the last axis, :296-297), pos_cumsum / neg_cumsum
 * float32 [n], coords_ids int64 [n], color_map uint8 [n, 3] (NULL with id_to_color_map NULL), successor int64 [n] or NULL (the end event),
 * mismatch int32 [1] or NULL: set to 1 when an end event is not on its start event's coordinate id (the reference asserts, :284), when an
 * id / successor / coordinate id (column 0, checked against n_coords = Ncoords) lies outside its table -- such an event gives zero outputs
 * --, or (track form) when an event's timestamp is not finite. */
int evd_sample_events(const double* events, long N, int ncol, const float* id_to_coords, long n_coords, const unsigned char* id_to_color_map,
                      const float* poses, const long long* events_ids, const long long* hops, long n, const float* K, int add_halfpix,
                      float* rays_start, float* rays_end, float* pos_cumsum, float* neg_cumsum, long long* coords_ids,
                      unsigned char* color_map, long long* successor, int* mismatch, void* stream);

/* ---------------------------------------------------------------- camera trajectory on the device
 * LLFFEventsDataset.interpolate_poses, data/loader_events.py:133-148: the pose of an event camera at a timestamp = scipy Slerp of the
 * key rotations + a cubic (not-a-knot) spline of the key translations (utils/data.py:34-62 _get_slerp_interpolator, built at
 * loader_events.py:175-182, queries clipped to the key range), the LLFF column change [r1, -r0, r2, t] (:137), float32, translation
 * x bd_scale (:140), then recenter_poses with the image dataset's average pose (:142-143, utils/data.py:167-183: inv(c2w) @ pose in
 * float64, stored float32).  The reference evaluates it on the CPU for every batch (scipy, .cpu().numpy() round trip at :280-283).
 * Here the HOST prepares, once per dataset, what does not depend on the query (evdeblurnerf_amd/poses.py, plain numpy): the key
 * rotations as unit quaternions (scipy's from_matrix: orthogonalised, then the largest-diagonal branch), the rotation vectors
 * log(q_i^-1 q_i+1) between neighbours, and the spline as one cubic per interval in u = (t - t_i) / (t_i+1 - t_i); the device evaluates
 * one pose per timestamp in float64 with the reference's float32 roundings at the same places.
 * --spherify (utils/data.py:189-252) is not covered: no shipped config sets it, and the reference's event path returns float64 there. */
typedef struct evd_pose_track {
    int n_keys;                 /* M >= 4 key poses (scipy's cubic interp1d needs 4) */
    const double* key_t;        /* dev [M] strictly ascending timestamps */
    const double* key_quat;     /* dev [M, 4] unit quaternions (x, y, z, w) of the key rotations */
    const double* key_rotvec;   /* dev [M-1, 3] rotation vector of q_i^-1 * q_i+1 */
    const double* trans_coef;   /* dev [M-1, 4, 3] c0..c3 of the translation on interval i, polynomial in u */
    float bd_scale;             /* loader_events.py:140 (float32 product, like the reference's in-place multiply) */
    int recenter;               /* non-zero: left-multiply by recenter_inv (loader_events.py:142-143) */
    double recenter_inv[12];    /* rows 0..2 of inv(c2w 4x4), row-major [3, 4] */
} evd_pose_track;
/* interpolate_poses(t): t dev float64 [n] -> poses dev float32 [n, 3, 4] (rows 0..2 of the reference's [n, 4, 4]; the fourth row is
 * 0 0 0 1) */
int evd_interpolate_poses(const evd_pose_track* track, const double* t, long n, float* poses, void* stream);
/* evd_sample_events with the start / end poses evaluated from the track at the events' own timestamps (column ncol-3) inside the
 * same launch, i.e. EventsDataset.sample_events as the reference runs it (loader_events.py:280-283) without a per-event pose table.
 * All other arguments as evd_sample_events. */
int evd_sample_events_track(const double* events, long N, int ncol, const float* id_to_coords, long n_coords, const unsigned char* id_to_color_map,
                            const evd_pose_track* track, const long long* events_ids, const long long* hops, long n, const float* K,
                            int add_halfpix, float* rays_start, float* rays_end, float* pos_cumsum, float* neg_cumsum, long long* coords_ids,
                            unsigned char* color_map, long long* successor, int* mismatch, void* stream);

/* ---------------------------------------------------------------- once-per-dataset event tables (last clause)
 * What LLFFEventsDataset.load_event_data (data/loader_events.py:150-257) and load_events_h5 (utils/events.py:11-69) compute on the CPU
 * from the ARRAYS of events.h5 (reading the file stays with the caller).  Data-dependent sizes are returned in device words; outputs are
 * sized for the worst case.  Mirror: evdeblurnerf_amd.events.EventTables.from_arrays.
 *
 * evd_event_coord_ids -- utils/events.py:39-66 with optimize_ids=True (what load_event_data passes, :186-188): the pixels no event
 *   rounds to (np.round, clipped; :39-42), then np.unique(return_index, return_inverse) over the float64 rows [event (x, y) ; silent
 *   pixel (x, y)] viewed as 16 raw bytes (utils/misc.py:143-149 to_flattenvoid): the ids follow the BYTE order of the little-endian pairs.
 *   x, y dev float32 [N]  ->  ev_coord_ids dev int64 [N], noev_coord_ids dev int64 [h w] (the first counts[1] valid, raster order of the
 *   silent pixels), id_to_coords dev float64 [N + h w, 2] (the first counts[0] rows valid), counts dev int64 [2] = (Ncoords, Nsilent).
 * evd_event_filter -- loader_events.py:191 (events with tmin <= t <= tmax, order kept) and :203-206 (if the smallest kept polarity is 0,
 *   zeros become -1): coord_ids int64 [N], t / p float64 [N] -> events_out dev float64 [N, 3] (id, t, p; the first *count rows valid),
 *   count dev int64 [1], bad_polarity dev int32 [1] or NULL: set when the kept polarities are not {-1, 1} afterwards (the reference asserts).
 * evd_event_color_map -- loader_events.py:208-236: id_to_color_map dev uint8 [Ncoords, 3], one-hot r / g / b of the Bayer pattern
 *   (r g / g b).  inv_mapx == NULL: integer coordinates, the colour of the pixel itself (:215-218).  Else inv_mapx / inv_mapy dev
 *   float32 [h, w] (ev_map.npz): the colour of the LAST pixel in raster order whose map entry equals the coordinate (:226-230; id_to_coords
 *   must be the table evd_event_coord_ids wrote -- it is searched in its byte order); unmapped dev int32 [1] or NULL: set when a coordinate
 *   that is not a silent pixel's (noev_coord_ids [n_noev]) got no colour (the reference asserts, :231-234). */
size_t evd_event_coord_ids_workspace_bytes(long N, int h, int w);
int evd_event_coord_ids(const float* x, const float* y, long N, int h, int w, long long* ev_coord_ids, long long* noev_coord_ids, double* id_to_coords,
                        long long* counts, void* workspace, size_t workspace_bytes, void* stream);
size_t evd_event_filter_workspace_bytes(long N);
int evd_event_filter(const long long* coord_ids, const double* t, const double* p, long N, double tmin, double tmax, double* events_out, long long* count,
                     int* bad_polarity, void* workspace, size_t workspace_bytes, void* stream);
size_t evd_event_color_map_workspace_bytes(long n_coords);
int evd_event_color_map(const double* id_to_coords, long n_coords, int h, int w, const float* inv_mapx, const float* inv_mapy, const long long* noev_coord_ids,
                        long n_noev, unsigned char* id_to_color_map, int* unmapped, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------- image batch assembly (first half)
 * LLFFDataset.__getitem__, data/loader.py:325-356: a batch of flat ray ids in [0, n_img * H * W) -> image id / pixel
 * (unravel_idx_from_rayid :118-123, C order), the image's pose, the pixel's colour (and the EDI prior's colour, set_pts0_prior :110-116),
 * the pixel-centre ray (get_rays_pix on the integer pixel with add_halfpix, utils/rays.py:25-36).  One launch on the resident dataset.
 *   ray_ids dev int64 [n]; images dev float32 [n_img, H, W, 3]; pts0_images the same shape or NULL; poses dev float32 [n_img, 3, 4];
 *   K host[9].
 * outputs (dev, the keys of the reference's dict): rays float32 [n, 3, 2] (origin, direction in the last axis), rays_x / rays_y
 * float32 [n] (pixel + 0.5), images_idx int64 [n], rgbsf float32 [n, 3], poses_out float32 [n, 3, 4], rgbsf_pts0 float32 [n, 3]
 * (NULL with pts0_images NULL), invalid int32 [1] or NULL: set to 1 when an id lies outside [0, n_img * H * W) (the reference raises
 * an IndexError; such a ray is written as zeros with images_idx -1). */
int evd_image_batch(const long long* ray_ids, long n, const float* images, const float* pts0_images, const float* poses, int n_img, int H,
                    int W, const float* K, float* rays, float* rays_x, float* rays_y, long long* images_idx, float* rgbsf, float* poses_out,
                    float* rgbsf_pts0, int* invalid, void* stream);

/* ---------------------------------------------------------------- measurement aid (no reference counterpart)
 * Sustained rate of back-to-back v_mfma_f32_32x32x16_bf16 issue on every SIMD of the current device, in dense
 * TFLOP/s, with constant (random_operands == 0) or random operands.  The chip clocks to its power budget, so the
 * second is the practical ceiling of any bf16/f16 MFMA kernel working on real data.  Synchronises the stream. */
int evd_probe_mfma_rate(int random_operands, int iters, double* tflops, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EVDNERF_H */
