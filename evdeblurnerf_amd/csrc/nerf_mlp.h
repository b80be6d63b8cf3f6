// Layout contract between the host-side weight packer (nerf_pack.cpp part of evd_api.hip) and the fused
// NeRF MLP kernel (kernel_nerf_mlp.hip).
//
// The kernel computes every layer TRANSPOSED:  D[out_feature, sample] = W[out, in] * X[in, sample]
//   A operand (32 x 16 per MFMA k-step) = a 32-row tile of the weight matrix       -> streamed through LDS
//   B operand (16 x 32)                 = the activations of 32 samples            -> live in registers
//   C/D (32 x 32, v_mfma_*_32x32x*)     : lane l, reg r holds D[(r&3) + 8(r>>2) + 4(l>>5)][l&31]
// so a lane owns ONE sample (n = l & 31) and the half h = l >> 5 of the features.  The D fragment of
// output tile mt is, after bias/activation/convert, directly the B fragment of the next layer's k-steps
// 2mt and 2mt+1 (k-step j <-> features 16j..16j+15) if the K index inside a k-step is permuted:
//      B position kk = 8h + e  (e = 0..7)   <->   feature 16j + phi(kk),  phi(kk) = 8((kk&7)>>2) + 4(kk>>3) + (kk&3)
// The packer applies the same permutation to the weight columns, so no LDS transpose or cross-lane
// shuffle is ever needed between layers.
//
// Positional-encoding inputs use a second arrangement chosen so both lane halves run the same code:
// position q = 8j + e holds, for q < 3L: frequency k = q/3, component c = q%3 -> h=0: sin(x_c 2^k), h=1: cos(x_c 2^k);
// q = 3L: (x_0 | x_1);  q = 3L+1: (x_2 | pad);  further positions: pad.
#pragma once

#include <cstddef>
#include <cstdint>

namespace evd {

constexpr int PE_L = 10;        // multires (reference options.py: --multires 10)
constexpr int PE_LV = 4;        // multires_views
constexpr int PE_KS = (3 * PE_L + 2 + 7) / 8;    // 4 k-steps hold the 63-wide point encoding
constexpr int PEV_KS = (3 * PE_LV + 2 + 7) / 8;  // 2 k-steps hold the 27-wide direction encoding
// Other frequency counts (options.py --multires / --multires_views) run in the generic kernels: the point encoding always takes PE_KS
// k-steps (multires <= PE_L_MAX, unused positions are zero columns of the packed weights), the direction encoding 2 or 4 k-steps.
constexpr int PE_L_MAX = (8 * PE_KS - 2) / 3;          // 10
__host__ __device__ constexpr int pe_ksteps(int L) { return (3 * L + 2 + 7) / 8; }

__host__ __device__ constexpr int phi(int kk) { return 8 * ((kk & 7) >> 2) + 4 * (kk >> 3) + (kk & 3); }
__host__ __device__ constexpr int phi_inv(int p) { return (p & 3) | ((p >> 3) << 2) | (((p >> 2) & 1) << 3); }

// source column of the reference PE vector ([x, sin(x f0), cos(x f0), ...], networks/embedding.py:88-98)
// for arrangement position (q, h); -1 = zero padding
__host__ __device__ constexpr int pe_src_col(int L, int q, int h) {
    return q < 3 * L ? 3 + 6 * (q / 3) + 3 * h + (q % 3) : (q == 3 * L ? h : (q == 3 * L + 1 ? (h == 0 ? 2 : -1) : -1));
}

constexpr bool is_half_prec(int prec) { return prec == 2 /*BF16*/ || prec == 3 /*F16*/; }   // one 2-byte operand per value
constexpr bool is_train_prec(int prec) { return is_half_prec(prec) || prec == 1 /*F16X3*/; }  // modes the training kernels are built for
constexpr int frag_bytes(int prec) { return is_half_prec(prec) ? 1024 : 2048; }
constexpr int mlp_threads(int prec) { return is_half_prec(prec) ? 512 : 256; }
constexpr int chunk_bytes(int prec) { return mlp_threads(prec) * 64; }
constexpr int frags_per_chunk(int prec) { return chunk_bytes(prec) / frag_bytes(prec); }

// tile-group size of the pipelined NeRF kernel = fragment order of its stream (groups of G tiles, k-steps, tiles innermost)
constexpr int nerf_group(int prec) { return 1; }

// kernel arguments of k_nerf_mlp
struct MlpParams {
    const char* wstream;    // packed fragment stream of the chosen precision
    const float* bias;      // 32 floats per output tile, stream order
    const float* ray_batch; // [R, ncol]
    const float* z;         // [R, S]
    long nsamp;             // R * S
    int S, ncol, D, skip, nchunks, nbias;
    float* raw;             // [R, S, 4]
    float* feature;         // [R, S, W] or null
    int feature_kind;       // 0 none, 1 after_linear, 2 before_linear
    char* act;              // training kernels: activation store (act_tile_bytes per 32-sample tile), else null
    int pe_l, pe_lv;        // multires / multires_views (generic kernels; the pipelined ones are built for PE_L / PE_LV)
    int no_views;           // use_viewdirs=False network: one output_linear head tile after the hidden layers, no view direction columns
    const unsigned* wscale; // compensated float16 mode: row-scale words, 32 per output tile in bias order (pack.h StreamBuilderC), else null
    // fused render step of the compensated float16 kernel (nerf_mlp_c_kernel.h, FUSE): z stratification in the prologue (renderer.py:163-178),
    // raw2outputs in the epilogue (nerf.py:74-129); z / raw / weights are written only where a pointer is given
    int fuse, lindisp, perturb, rgb_act, sigma_act, white_bkgd;
    const float* t_rand;    // [R, S] or null
    float *z_out, *rgb_map, *depth_map, *acc_map, *weights;
};

// Activation / gradient store of the training path, in 1 KiB fragments per 32-sample tile (W = 256, D = 8):
// what the forward saves for the backward kernels, then the gradient fragments the dgrad chain hands to wgrad.
namespace astore {
constexpr int PE = 0, DIR = 4, H0 = 6;                    // H_l at H0 + 16 l: the ReLU output of pts_linears[l]
constexpr int F = H0 + 16 * 8, HV = F + 16, FWD_END = HV + 8;   // feature_linear output, views hidden
constexpr int G_RGB = FWD_END, G_ALPHA = G_RGB + 1;       // d raw as two single-k-step fragments
constexpr int D_HV = G_ALPHA + 1, D_F = D_HV + 8, D_H0 = D_F + 16;    // d loss / d pre-activation of hv, feature, h_l (at D_H0 + 16 l)
constexpr int D_PE0 = D_H0 + 16 * 8, D_PE5 = D_PE0 + PE_KS, D_DIRG = D_PE5 + PE_KS;   // d PE(pts) via pts_linears[0] / the skip layer, d PE(dirs): the encodings' own arrangement
// ReLU patterns as bit masks (1 bit per activation: byte j of a lane's 16 bytes = the 8 elements of fragment j): what the dgrad
// epilogue reads instead of the 16 activation fragments
constexpr int M_H0 = D_DIRG + PEV_KS, M_HV = M_H0 + 8;
constexpr int TILE_FRAGS = M_HV + 1;
constexpr long TILE_BYTES = (long)TILE_FRAGS * 1024;      // the single-product half-precision modes
// bytes of a tile in arithmetic mode `prec`: the split-float16 mode keeps every fragment as a (hi, lo) pair in a 2 KiB slot
constexpr long tile_bytes(int prec) { return (long)TILE_FRAGS * frag_bytes(prec); }
}  // namespace astore

}  // namespace evd
