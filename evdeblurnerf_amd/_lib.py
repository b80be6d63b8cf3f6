"""ctypes binding of libevdnerf.so (include/evdnerf.h).

The library is the product: there is no CPU or PyTorch fallback. If the shared
object is missing or a call fails this module raises, loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EVD_LIB_PATH") or os.path.join(_HERE, "lib", "libevdnerf.so")
MAXL = 16

PREC = {"f32": 0, "f16x3": 1, "bf16": 2, "f16": 3, "f16c": 4, "f16m": 5}     # f16m: training entries only (include/evdnerf.h)
ACT = {"none": 0, "relu": 1, "sigmoid": 2, "exp": 3, "sigmoid1": 4, "softplus": 5, "tanh": 6}

_fp = C.POINTER(C.c_float)
_vp = C.c_void_p


class EvdError(RuntimeError):
    pass


class RenderCfg(C.Structure):
    _fields_ = [("H", C.c_int), ("W", C.c_int), ("focal", C.c_float),
                ("ndc", C.c_int), ("use_viewdirs", C.c_int), ("lindisp", C.c_int), ("N_samples", C.c_int),
                ("N_importance", C.c_int), ("white_bkgd", C.c_int),
                ("near", C.c_float), ("far", C.c_float), ("perturb", C.c_float),
                ("is_train", C.c_int), ("precision", C.c_int)]


class NerfDesc(C.Structure):
    _fields_ = [("D", C.c_int), ("W", C.c_int), ("multires", C.c_int), ("multires_views", C.c_int), ("skip", C.c_int),
                ("rgb_act", C.c_int), ("sigma_act", C.c_int), ("rmnear", C.c_float),
                ("pts_w", _fp * MAXL), ("pts_b", _fp * MAXL),
                ("views_w", _fp), ("views_b", _fp), ("feature_w", _fp), ("feature_b", _fp),
                ("alpha_w", _fp), ("alpha_b", _fp), ("rgb_w", _fp), ("rgb_b", _fp),
                ("output_w", _fp), ("output_b", _fp), ("output_ch", C.c_int)]


class NerfGrads(C.Structure):
    _fields_ = [("pts_w", _vp * MAXL), ("pts_b", _vp * MAXL),
                ("views_w", _vp), ("views_b", _vp), ("feature_w", _vp), ("feature_b", _vp),
                ("alpha_w", _vp), ("alpha_b", _vp), ("rgb_w", _vp), ("rgb_b", _vp), ("accumulate", C.c_int)]


class VoxelGrads(C.Structure):
    _fields_ = [("sigma_w", _vp * 2), ("color_w", _vp * 3), ("color_b", _vp * 3), ("accumulate", C.c_int)]


class AwpEmbedGrads(C.Structure):
    _fields_ = [("w", _vp * 4), ("b", _vp * 4)]


class AwpTailDesc(C.Structure):
    _fields_ = [("P", C.c_int), ("S", C.c_int), ("VF", C.c_int), ("dir_freqs", C.c_int), ("n_mot", C.c_int), ("training", C.c_int),
                ("bn_eps", C.c_float), ("bn_momentum", C.c_float)]


class VoxelGridGrads(C.Structure):
    _fields_ = [("plane", _vp * 3), ("line", _vp * 3), ("basis", _vp)]


class VoxelDesc(C.Structure):
    _fields_ = [("num_layers", C.c_int), ("hidden_dim", C.c_int), ("geo_feat_dim", C.c_int),
                ("num_layers_color", C.c_int), ("input_ch", C.c_int), ("multires", C.c_int), ("multires_views", C.c_int),
                ("app_dim", C.c_int), ("n_comp", C.c_int * 3), ("grid", C.c_int * 3), ("app_act", C.c_int),
                ("rgb_act", C.c_int), ("sigma_act", C.c_int), ("composite_feature", C.c_int),
                ("aabb", C.c_float * 6), ("rmnear", C.c_float),
                ("sigma_w", _fp * MAXL), ("color_w", _fp * MAXL), ("color_b", _fp * MAXL),
                ("plane", _fp * 3), ("line", _fp * 3), ("basis", _fp)]


class CrfDesc(C.Structure):
    _fields_ = [("map_type", C.c_int), ("gamma", C.c_float), ("extra_features", C.c_int),
                ("w", _fp * 4), ("b", _fp * 4)]


class PoseTrack(C.Structure):
    _fields_ = [("n_keys", C.c_int), ("key_t", _vp), ("key_quat", _vp), ("key_rotvec", _vp), ("trans_coef", _vp),
                ("bd_scale", C.c_float), ("recenter", C.c_int), ("recenter_inv", C.c_double * 12)]


class RenderOut(C.Structure):
    _fields_ = [(k, _vp) for k in ("rgb", "depth", "acc", "z_vals", "weights", "rgb0", "depth0", "acc0", "z_std",
                                   "z_vals0", "weights0", "feature", "raw")] + [("feature_kind", C.c_int)]


# name -> (restype, argtypes); every symbol include/evdnerf.h declares
_L, _I, _F, _S = C.c_long, C.c_int, C.c_float, C.c_size_t
SIGNATURES = {
    "evd_last_error": (C.c_char_p, []),
    "evd_version": (_I, []),
    "evd_debug_side_spin_count": (_L, []),
    "evd_compute_successor_workspace_bytes": (_S, [_L]),
    "evd_compute_successor": (_I, [_vp, _L, _L, _vp, _vp, _vp, _vp, _vp, _S, _vp]),
    "evd_event_coord_ids_workspace_bytes": (_S, [_L, _I, _I]),
    "evd_event_coord_ids": (_I, [_vp, _vp, _L, _I, _I, _vp, _vp, _vp, _vp, _vp, _S, _vp]),
    "evd_event_filter_workspace_bytes": (_S, [_L]),
    "evd_event_filter": (_I, [_vp, _vp, _vp, _L, C.c_double, C.c_double, _vp, _vp, _vp, _vp, _S, _vp]),
    "evd_event_color_map_workspace_bytes": (_S, [_L]),
    "evd_event_color_map": (_I, [_vp, _L, _I, _I, _vp, _vp, _vp, _L, _vp, _vp, _vp, _S, _vp]),
    "evd_sample_events": (_I, [_vp, _L, _I, _vp, _L, _vp, _vp, _vp, _vp, _L, _fp, _I, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "evd_sample_events_track": (_I, [_vp, _L, _I, _vp, _L, _vp, C.POINTER(PoseTrack), _vp, _vp, _L, _fp, _I, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "evd_interpolate_poses": (_I, [C.POINTER(PoseTrack), _vp, _L, _vp, _vp]),
    "evd_image_batch": (_I, [_vp, _L, _vp, _vp, _vp, _I, _I, _I, _fp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "evd_rbk_warp": (_I, [_vp, _vp, _vp, _L, _I, _I, _vp, _vp, _vp]),
    "evd_awp_feature_integration": (_I, [_vp, _vp, _vp, _L, _I, _I, _vp, _vp]),
    "evd_awp_feature_integration_bwd": (_I, [_vp, _vp, _vp, _vp, _L, _I, _I, _vp, _vp, _vp, _vp]),
    "evd_mam_local_forward": (_I, [_vp, _vp, _L, _I, _I, _I, _vp, _vp, _vp, _vp, _vp]),
    "evd_mam_local_backward": (_I, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _L, _I, _I, _I, _vp, _vp, _I, _vp, _vp]),
    "evd_awp_tail_num_params": (_I, [_I]),
    "evd_awp_tail_param_count": (_L, [C.POINTER(AwpTailDesc)]),
    "evd_awp_tail_workspace_bytes": (_S, [C.POINTER(AwpTailDesc), _L, _I]),
    "evd_awp_tail_saved_floats": (_L, [C.POINTER(AwpTailDesc)]),
    "evd_awp_tail_forward": (_I, [C.POINTER(AwpTailDesc), C.POINTER(_vp), _vp, _vp, _vp, _vp, _vp, _L, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _S, _vp]),
    "evd_awp_tail_backward": (_I, [C.POINTER(AwpTailDesc), C.POINTER(_vp), _vp, _vp, _vp, _vp, _vp, _L, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                   _vp, _vp, _vp, _S, _vp]),
    "evd_awp_local_consumers_backward": (_I, [_vp] * 11 + [_L, _I, _I, _I, _vp, _vp, _vp, _vp, _vp, _vp]),
    "evd_awp_embed_create": (_I, [C.POINTER(_fp), C.POINTER(_fp), _I, _I, _I, C.POINTER(_vp)]),
    "evd_awp_embed_destroy": (None, [_vp]),
    "evd_awp_embed_param_count": (_L, [_vp]),
    "evd_awp_embed_load_params": (_I, [_vp, _vp, _vp]),
    "evd_awp_embed_store_bytes": (_S, [_vp, _L]),
    "evd_awp_embed_backward_workspace_bytes": (_S, []),
    "evd_awp_embed_forward": (_I, [_vp, _I, _vp, _vp, _vp, _S, _L, _vp, _vp, _S, _vp]),
    "evd_awp_embed_backward": (_I, [_vp, _I, _vp, _L, _vp, _S, C.POINTER(AwpEmbedGrads), _vp, _vp, _vp, _S, _vp]),
    "evd_probe_mfma_rate": (_I, [_I, _I, C.POINTER(C.c_double), _vp]),
    "evd_device_count": (_I, []),
    "evd_get_rays": (_I, [_I, _I, _fp, _fp, _I, _vp, _vp, _vp]),
    "evd_get_rays_pix": (_I, [_vp, _fp, _vp, _L, _I, _vp, _vp, _vp]),
    "evd_ndc_rays": (_I, [_I, _I, _F, _F, _vp, _vp, _L, _vp, _vp, _vp]),
    "evd_embed": (_I, [_vp, _L, _I, _I, _vp, _vp]),
    "evd_ray_batch": (_I, [C.POINTER(RenderCfg), _vp, _L, _vp, _vp]),
    "evd_ray_batch_bwd": (_I, [C.POINTER(RenderCfg), _vp, _vp, _L, _vp, _vp]),
    "evd_points": (_I, [_vp, _I, _vp, _L, _I, _vp, _vp]),
    "evd_points_bwd": (_I, [_vp, _vp, _L, _I, _I, _vp, _vp]),
    "evd_sample_z": (_I, [C.POINTER(RenderCfg), _vp, _I, _L, _vp, _vp, _vp]),
    "evd_nerf_create": (_I, [C.POINTER(NerfDesc), C.POINTER(_vp)]),
    "evd_nerf_destroy": (None, [_vp]),
    "evd_nerf_param_count": (_L, [_vp]),
    "evd_nerf_param_blocks": (_I, [_vp, C.POINTER(C.c_long), _I]),
    "evd_nerf_load_params": (_I, [_vp, _vp, _vp]),
    "evd_nerf_stream_bytes": (_S, [_vp, _I]),
    "evd_nerf_mlp": (_I, [_vp, _I, _vp, _vp, _L, _I, _vp, _vp, _I, _vp]),
    "evd_nerf_train_store_bytes": (_S, [_L]),
    "evd_nerf_train_store_bytes_prec": (_S, [_I, _L]),
    "evd_nerf_mlp_train": (_I, [_vp, _I, _vp, _vp, _L, _I, _vp, _vp, _S, _vp]),
    "evd_nerf_backward_workspace_bytes": (_S, []),
    "evd_nerf_mlp_backward": (_I, [_vp, _I, _vp, _L, _I, _vp, _S, C.POINTER(NerfGrads), _vp, _vp, _I, _vp, _vp, _vp, _S, _vp]),
    "evd_voxel_param_count": (_L, [_vp]),
    "evd_voxel_param_blocks": (_I, [_vp, C.POINTER(C.c_long), _I]),
    "evd_voxel_load_params": (_I, [_vp, _vp, _vp]),
    "evd_voxel_train_store_bytes": (_S, [_vp, _L]),
    "evd_voxel_train_store_bytes_prec": (_S, [_vp, _I, _L]),
    "evd_voxel_backward_workspace_bytes": (_S, []),
    "evd_voxel_mlp_train": (_I, [_vp, _I, _vp, _vp, _I, _vp, _I, _L, _I, _vp, _vp, _vp, _S, _vp]),
    "evd_voxel_geo_feat_dim": (_I, [_vp]),
    "evd_voxel_mlp_backward": (_I, [_vp, _I, _vp, _vp, _vp, _vp, _S, _L, _I, _vp, _S, C.POINTER(VoxelGrads), _vp, _I, _vp, _vp, _I, _vp, _vp, _vp, _S, _vp]),
    "evd_voxel_grid_sizes": (_I, [_vp, C.POINTER(C.c_long)]),
    "evd_voxel_get_grids": (_I, [_vp, C.POINTER(_vp), C.POINTER(_vp), _vp, _vp]),
    "evd_voxel_load_grids": (_I, [_vp, C.POINTER(_vp), C.POINTER(_vp), _vp, _vp]),
    "evd_voxel_sample_bwd": (_I, [_vp, _vp, _L, _vp, _I, _I, C.POINTER(VoxelGridGrads), _vp, _vp]),
    "evd_voxel_sample_bwd_workspace_bytes": (_S, [_vp, _L]),
    "evd_voxel_sample_bwd_ws": (_I, [_vp, _vp, _L, _vp, _I, _I, C.POINTER(VoxelGridGrads), _vp, _vp, _S, _vp]),
    "evd_voxel_sample_bwd_prec": (_I, [_vp, _I, _vp, _L, _vp, _I, _I, C.POINTER(VoxelGridGrads), _vp, _vp, _S, _vp]),
    "evd_voxel_tv_loss_bwd": (_I, [_vp, _vp, C.POINTER(VoxelGridGrads), _vp]),
    "evd_raw2outputs": (_I, [_vp, _vp, _vp, _I, _L, _I, _I, _I, _I, _I, _I, _I, _I, _F, _vp,
                             _vp, _vp, _vp, _vp, _vp, _vp, _I, _vp, _vp]),
    "evd_raw2outputs_bwd": (_I, [_vp, _vp, _vp, _I, _L, _I, _I, _I, _I, _I, _I, _I, _I, _F, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "evd_raw2outputs_bwd_rays": (_I, [_vp, _vp, _vp, _I, _L, _I, _I, _I, _I, _I, _I, _I, _I, _F, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _I, _vp]),
    "evd_sample_pdf_merge": (_I, [_vp, _vp, _L, _I, _I, _I, _vp, _vp, _vp, _vp, _vp, _vp]),
    "evd_nerf_render_workspace_bytes": (_S, [C.POINTER(RenderCfg), _L]),
    "evd_nerf_render_rays": (_I, [_vp, _vp, C.POINTER(RenderCfg), _vp, _L, _vp, _vp, _vp, _vp,
                                  C.POINTER(RenderOut), _vp, _S, _vp]),
    "evd_nerf_render": (_I, [_vp, _vp, C.POINTER(RenderCfg), _vp, _L, _vp, _vp, _vp, _vp,
                             C.POINTER(RenderOut), _vp, _S, _vp]),
    "evd_voxel_create": (_I, [C.POINTER(VoxelDesc), C.POINTER(_vp)]),
    "evd_voxel_destroy": (None, [_vp]),
    "evd_voxel_sample": (_I, [_vp, _vp, _L, _vp, _I, _I, _vp]),
    "evd_voxel_sample_prec": (_I, [_vp, _I, _vp, _L, _vp, _I, _I, _vp]),
    "evd_voxel_forward": (_I, [_vp, _I, _vp, _vp, _I, _vp, _I, _vp, _vp, _I, _L, _I, _I,
                               _vp, _vp, _vp, _vp, _vp, _vp, _S, _vp]),
    "evd_voxel_forward_workspace_bytes": (_S, [_vp, _L, _I]),
    "evd_c2f_render_workspace_bytes": (_S, [_vp, _vp, C.POINTER(RenderCfg), _L]),
    "evd_c2f_render_rays": (_I, [_vp, _vp, C.POINTER(RenderCfg), _vp, _L, _vp, _vp, _vp, _vp,
                                 C.POINTER(RenderOut), _vp, _S, _vp]),
    "evd_c2f_render": (_I, [_vp, _vp, C.POINTER(RenderCfg), _vp, _L, _vp, _vp, _vp, _vp,
                            C.POINTER(RenderOut), _vp, _S, _vp]),
    "evd_voxel_tv_loss": (_I, [_vp, _vp, _vp]),
    "evd_merge_features": (_I, [_vp, _vp, _vp, _L, _I, _I, _I, _vp, _I, _vp]),
    "evd_merge_features_bwd": (_I, [_vp, _I, _vp, _L, _I, _I, _I, _vp, _vp, _vp]),
    "evd_weighted_sum": (_I, [_vp, _vp, _L, _I, _I, _vp, _vp]),
    "evd_crf_create": (_I, [C.POINTER(CrfDesc), C.POINTER(_vp)]),
    "evd_crf_destroy": (None, [_vp]),
    "evd_crf_forward": (_I, [_vp, _vp, _vp, _I, _I, _I, _L, _vp, _vp]),
    "evd_blur_loss_reduce": (_I, [_vp, _I, _vp, _vp, _vp, _vp, _vp, _vp, _L, _I, _vp, _vp, _vp, _vp, _vp]),
    "evd_crf_param_count": (_I, []),
    "evd_crf_get_params": (_I, [_vp, _vp]),
    "evd_crf_load_params": (_I, [_vp, _vp]),
    "evd_event_loss_bwd": (_I, [_vp, _I, _I, _I, _vp, _vp, _vp, _vp, _vp, _vp, _F, _F, _vp, _fp, _L, _F, _F, _vp, _vp, _vp, _vp, _vp, _vp]),
    "evd_blur_loss_bwd": (_I, [_vp, _I, _vp, _vp, _vp, _vp, _vp, _vp, _L, _I, _fp, _vp, _vp, _vp, _vp, _vp]),
    "evd_blur_loss_bwd_dev": (_I, [_vp, _I, _vp, _vp, _vp, _vp, _vp, _vp, _L, _I, _vp, _vp, _vp, _vp, _vp, _vp]),
    "evd_event_loss_bwd_dev": (_I, [_vp, _I, _I, _I, _vp, _vp, _vp, _vp, _vp, _vp, _F, _F, _vp, _fp, _L, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "evd_event_loss_reduce": (_I, [_vp, _I, _I, _I, _vp, _vp, _vp, _vp, _vp, _vp, _F, _F, _vp, _fp, _L, _vp, _vp]),
    "evd_numerics_flags": (_I, [C.POINTER(_vp), C.POINTER(C.c_long), _I, _vp, _vp]),
    "evd_edi_deblur": (_I, [_vp, _vp, _I, _L, _vp, _vp]),
    "evd_edi_bii_image": (_I, [_vp, _vp, _vp, _L, _I, _I, _F, _F, _vp, _vp]),
}

_lib = None


def _share_torch_hip_runtime():
    """libevdnerf.so takes torch's device pointers and HIP streams, so it must run on the SAME HIP runtime
    instance as torch. PyTorch-ROCm bundles its own libamdhip64.so.7 (same SONAME as /opt/rocm's): load torch's
    copy into the process first, then our DT_NEEDED entry resolves to it instead of a second runtime."""
    import torch  # noqa: F401  (pulls in torch/lib/libamdhip64.so when built for ROCm)
    tl = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(tl):
        C.CDLL(tl, mode=C.RTLD_GLOBAL)


def lib():
    """The loaded library; raises EvdError if it is missing (no fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EvdError(f"{LIB_PATH} not found: build it with `python -m evdeblurnerf_amd.build` "
                           "(hipcc, --offload-arch=gfx950). There is no CPU fallback.")
        _share_torch_hip_runtime()
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)   # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().evd_last_error()
        raise EvdError(f"{what or 'libevdnerf'} failed ({rc}): {msg.decode() if msg else '?'}")


def ptr(t):
    """Device pointer of a contiguous float32/int32/uint8 torch tensor (or None)."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise EvdError("tensor must be contiguous")
    return C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# Generation counter of the training path: every backward of the library's autograd nodes bumps it, and the parameter / grid re-pack checks
# (nerf.py, voxnerf.py) treat "a backward ran since the last re-pack" like a changed tensor version.  tensor._version alone is not enough:
# torch.optim.Adam(fused=True) updates the parameters WITHOUT bumping their version counters, and a library that went on rendering with
# its stale packed copies would train nothing, silently.
_BACKWARD_GEN = [0]


def backward_generation():
    return _BACKWARD_GEN[0]


def note_backward():
    _BACKWARD_GEN[0] += 1

