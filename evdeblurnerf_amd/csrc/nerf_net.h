// The opaque evd_nerf handle of include/evdnerf.h, shared by evd_api.hip and evd_train_api.hip.
#pragma once

#include "evd_common.h"
#include "pack.h"

// transposed-weight streams of the dgrad chain (nerf_train_kernel.h), in the order the chain runs them
enum { EVD_BWD_RGB = 0, EVD_BWD_VIEWS, EVD_BWD_HEAD, EVD_BWD_PE0, EVD_BWD_PESKIP, EVD_BWD_DIR, EVD_BWD_HIDDEN1, EVD_BWD_NSTREAMS = EVD_BWD_HIDDEN1 + EVD_MAX_LAYERS };

struct evd_nerf {
    typedef evd::PackedStream Packed;
    int D, W, skip, rgb_act, sigma_act;
    int multires = 10, multires_views = 4;   // frequency counts of the two encodings (others than 10 / 4: generic kernel only)
    int no_views = 0;                        // use_viewdirs=False: output_linear head, 8-column ray batch, generic kernel only
    float rmnear;
    Packed stream[EVD_NUM_PREC];             // generic kernel: fragment streams per precision
    int nchunks[EVD_NUM_PREC];
    Packed pipe[EVD_NUM_PREC];               // software-pipelined kernel (where built): its own fragment order and chunking
    int pipe_chunks[EVD_NUM_PREC];
    evd::DevBuf bias, bias_src;
    evd::PackedStreamC pipe_c;               // EVD_PREC_F16C: float16 + fp6 fragment stream, row-scale words (32 per output tile in bias order), re-pack maps
    Packed bwd[EVD_NUM_PREC][EVD_BWD_NSTREAMS];   // training (bf16 / f16, 8 x 256): W^T streams; HIDDEN1 + l - 1 = pts_linears[l]
    evd::DevBuf wmaps;                       // wgrad index maps (int32), nerf_train.h
    int nparam_blocks;                       // 2 D + 8 parameter tensors, canonical order (evd_api.hip: nerf_param_sizes)
    long param_off[2 * EVD_MAX_LAYERS + 9];  // arena offset of each, [nparam_blocks] = total
    evd::RepackBatch batch;                  // table of every fragment stream, for the one-launch re-pack of evd_nerf_load_params
    mutable evd::SideStream side;            // backward entry: wgrad side stream of THIS handle (created on first use)
};
