// Wave-level DPP primitives and the fast activations of the compositing scan, shared by kernels_render.hip (k_composite_rows) and the
// fused step of the compensated-float16 NeRF kernel (nerf_mlp_c_kernel.h).
#pragma once

#include "evd_common.h"

namespace evd {

// ------------------------------------------------------------------------------------------------
// DPP (data-parallel primitive) wave operations: register-to-register lane exchange inside the VALU, no LDS
// crossbar round trip (which is what __shfl / ds_bpermute costs).  dpp_ctrl codes: quad_perm 0x00-0xFF,
// row_shr:n 0x110+n, wave_shl:1 0x130, wave_shr:1 0x138, row_mirror 0x140, row_half_mirror 0x141,
// row_bcast:15 0x142, row_bcast:31 0x143 (gfx9 family).
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_f32(float old, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL, ROW_MASK, 0xf, false));
}
// inclusive product scan over the 64 lanes (Kogge-Stone inside rows of 16, then row broadcasts)
__device__ __forceinline__ float wave_scan_mul_dpp(float v) {
    v *= dpp_f32<0x111>(1.f, v);
    v *= dpp_f32<0x112>(1.f, v);
    v *= dpp_f32<0x114>(1.f, v);
    v *= dpp_f32<0x118>(1.f, v);
    v *= dpp_f32<0x142, 0xa>(1.f, v);     // rows 1, 3 <- lane 15 of the row below
    v *= dpp_f32<0x143, 0xc>(1.f, v);     // rows 2, 3 <- lane 31
    return v;
}
// sum over the 64 lanes, result uniform (in an SGPR-backed value)
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += dpp_f32<0xb1>(0.f, v);           // quad_perm [1,0,3,2]
    v += dpp_f32<0x4e>(0.f, v);           // quad_perm [2,3,0,1]
    v += dpp_f32<0x141>(0.f, v);          // row_half_mirror
    v += dpp_f32<0x140>(0.f, v);          // row_mirror: every lane now holds its row's sum
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
}

// activations of the bandwidth-form scan: sigmoid with the hardware exp2 and reciprocal (relative error ~3e-7; the
// IEEE expf + division of evd::act() made the scan VALU-bound: 3.8 instead of 5.2 TB/s)
__device__ __forceinline__ float act_fast(int code, float x) {
    if (code == EVD_ACT_SIGMOID) return __builtin_amdgcn_rcpf(1.f + __expf(-x));
    if (code == EVD_ACT_RELU) return fmaxf(x, 0.f);
    if (code == EVD_ACT_NONE) return x;
    return act(code, x);
}


}  // namespace evd
