"""Seed-derived synthetic parameters, keyed by the reference's state_dict names.

Both sides of every parity test (the imported reference when goldens are
generated, the C oracle and the HIP library at test time) derive parameters
from ``numpy.random.RandomState(seed)`` in the fixed order below, so no weight
blobs are committed. The names are those ``named_parameters()`` yields for the
reference modules (reference networks/nerf.py:23-44,
networks/pdrf/voxnerf.py:47-105, networks/tonemapping.py:16-22); a real
checkpoint (run_nerf.py:617-638 ``network_state_dict``) uses the same names,
which is why ``pack.py`` consumes dicts of this shape.

Pure numpy: no torch, no reference import.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np


def pe_dim(multires: int, input_dim: int = 3) -> int:
    """Output width of the positional encoder (reference networks/embedding.py:65-86)."""
    return input_dim * (1 + 2 * multires)


def _uniform(rs: np.random.RandomState, shape, bound: float) -> np.ndarray:
    return rs.uniform(-bound, bound, size=shape).astype(np.float32)


def _linear(rs, sd, name, fan_in, fan_out, bias=True):
    bound = 1.0 / math.sqrt(fan_in)
    sd[f"{name}.weight"] = _uniform(rs, (fan_out, fan_in), bound)
    if bias:
        sd[f"{name}.bias"] = _uniform(rs, (fan_out,), bound)


def make_nerf_state_dict(seed: int, D: int = 8, W: int = 256, input_ch: int = 63,
                         input_ch_views: int = 27, skips=(4,), use_viewdirs: bool = True,
                         rgb_add_bias: bool = True, output_ch: int = 4,
                         sigma_bias_shift: float = 1.0) -> "OrderedDict[str, np.ndarray]":
    """Parameters of one reference ``NeRF`` module (networks/nerf.py:8-44).

    ``sigma_bias_shift`` is added to ``alpha_linear.bias`` so that densities of a
    random-init net are not all clamped to zero by the relu.
    """
    rs = np.random.RandomState(seed)
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for i in range(D):
        if i == 0:
            fan_in = input_ch
        elif (i - 1) in skips:
            fan_in = W + input_ch
        else:
            fan_in = W
        _linear(rs, sd, f"pts_linears.{i}", fan_in, W)
    _linear(rs, sd, "views_linears.0", input_ch_views + W, W // 2)
    if use_viewdirs:
        _linear(rs, sd, "feature_linear", W, W)
        _linear(rs, sd, "alpha_linear", W, 1)
        sd["alpha_linear.bias"] = (sd["alpha_linear.bias"] + np.float32(sigma_bias_shift)).astype(np.float32)
        _linear(rs, sd, "rgb_linear", W // 2, 3, bias=rgb_add_bias)
    else:
        _linear(rs, sd, "output_linear", W, output_ch)
        b = sd["output_linear.bias"].copy()
        b[3] += np.float32(sigma_bias_shift)                       # channel 3 is the density (nerf.py:89,99)
        sd["output_linear.bias"] = b.astype(np.float32)
    return sd


def pdrf_grid_size(aabb_min, aabb_max, n_voxels: int):
    """Grid resolution rule of reference networks/pdrf/voxnerf.py:88-93 (float32 maths like torch)."""
    lo = np.asarray(aabb_min, dtype=np.float32)
    hi = np.asarray(aabb_max, dtype=np.float32)
    ext = (hi - lo).astype(np.float32)
    voxel = np.float32(np.power(np.float32(np.prod(ext, dtype=np.float32) / np.float32(n_voxels)),
                                np.float32(1.0 / 3.0)))
    return [int(v) for v in (ext / voxel).astype(np.int64)]


def make_pdrf_state_dict(seed: int, grid_size, input_ch: int, input_ch_views: int = 27,
                         num_layers: int = 2, hidden_dim: int = 64, geo_feat_dim: int = 15,
                         num_layers_color: int = 3, app_dim: int = 32, app_n_comp=(64, 16, 16),
                         add_bias_color: bool = False, grid_scale: float = 0.1):
    """Parameters of one reference ``VoxelNeRFBase`` (networks/pdrf/voxnerf.py:47-122).

    ``grid_size`` is [gx, gy, gz]; plane i is [1, C_i, grid[mat1], grid[mat0]] with
    matMode [[0,1],[0,2],[1,2]], line i is [1, C_i, grid[vec], 1] with vecMode [2,1,0].
    The colour net hidden width is ``hidden_dim`` (reference quirk, voxnerf.py:73,78).
    """
    rs = np.random.RandomState(seed)
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for l in range(num_layers):
        fan_in = input_ch if l == 0 else hidden_dim
        fan_out = 1 + geo_feat_dim if l == num_layers - 1 else hidden_dim
        _linear(rs, sd, f"sigma_net.{l}", fan_in, fan_out, bias=False)
    for l in range(num_layers_color):
        fan_in = input_ch_views + geo_feat_dim if l == 0 else hidden_dim
        fan_out = 3 if l == num_layers_color - 1 else hidden_dim
        _linear(rs, sd, f"color_net.{l}", fan_in, fan_out, bias=add_bias_color)
    mat_mode = [[0, 1], [0, 2], [1, 2]]
    vec_mode = [2, 1, 0]
    for i in range(3):
        m0, m1 = mat_mode[i]
        sd[f"app_plane.{i}"] = (grid_scale * rs.standard_normal(
            (1, app_n_comp[i], grid_size[m1], grid_size[m0]))).astype(np.float32)
    for i in range(3):
        sd[f"app_line.{i}"] = (grid_scale * rs.standard_normal(
            (1, app_n_comp[i], grid_size[vec_mode[i]], 1))).astype(np.float32)
    _linear(rs, sd, "basis_mat", sum(app_n_comp), app_dim, bias=False)
    return sd


def make_crf_state_dict(seed: int, extra_features: int = 0):
    """Parameters of the reference learnable ``CRF`` MLP (networks/tonemapping.py:16-22)."""
    rs = np.random.RandomState(seed)
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    dims = [(1 + extra_features, 16), (16, 16), (16, 16), (16, 1)]
    for idx, (fi, fo) in zip((0, 2, 4, 6), dims):
        _linear(rs, sd, f"linear.{idx}", fi, fo)
    return sd


def make_awp_embed_state_dict(seed: int, input_ch: int = 128, W_sam: int = 64, D_sam: int = 4):
    """``sample_feature_embed_layer.{l}.{weight,bias}`` of the reference's AdaptiveWeightProposal (networks/dpnerf/awp.py:36-37)."""
    rs = np.random.RandomState(seed)
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for l in range(D_sam):
        _linear(rs, sd, f"sample_feature_embed_layer.{l}", input_ch if l == 0 else W_sam, W_sam)
    return sd


def prefixed(sd, prefix: str):
    """``{prefix}.{name}`` view of a state dict (e.g. ``mlp_coarse``)."""
    return OrderedDict((f"{prefix}.{k}", v) for k, v in sd.items())


def make_train_call_state_dict(seed: int, grid_coarse, grid_fine):
    """Both PDRF levels of the training-call goldens G32 / G33 (tools/gen_golden.py ``_train_call_model``): the default initialisation
    with denser fields (sigma_net.1 x 2: weights that vary along the ray and importance samples that move, while sample_pdf's
    (u - cdf) / pdf stays conditioned well enough that reference and kernels agree to 1e-5 on every pixel -- at x 3 a third of the
    rays carry a sample that moves by 4e-4 between two float32 implementations and a colour by 7e-5) and colours away from sigmoid(0)
    (color_net.2 x 8: the P sub-exposure rays of a pixel differ by a few 1e-2); rgb_add_bias off as in every shipped config."""
    sd = prefixed(make_pdrf_state_dict(seed * 10 + 3, grid_coarse, input_ch=32 + 63, hidden_dim=64, geo_feat_dim=15), "mlp_coarse")
    sd.update(prefixed(make_pdrf_state_dict(seed * 10 + 4, grid_fine, input_ch=64 + 63, hidden_dim=256, geo_feat_dim=128), "mlp_fine"))
    for k in list(sd):
        if k.endswith("sigma_net.1.weight"):
            sd[k] = sd[k] * np.float32(2.0)
        elif k.endswith("color_net.2.weight"):
            sd[k] = sd[k] * np.float32(8.0)
    return sd


# ---------------------------------------------------------------------------
# synthetic LLFF-shaped inputs
# ---------------------------------------------------------------------------

def synthetic_camera(H: int = 400, W: int = 400, focal: float = 400.0) -> np.ndarray:
    return np.array([[focal, 0.0, 0.5 * W], [0.0, focal, 0.5 * H], [0.0, 0.0, 1.0]], dtype=np.float32)


def _small_rotation(rs: np.random.RandomState, max_deg: float) -> np.ndarray:
    ang = np.deg2rad(rs.uniform(-max_deg, max_deg, size=3))
    cx, cy, cz = np.cos(ang)
    sx, sy, sz = np.sin(ang)
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return (rz @ ry @ rx)


def synthetic_pose(seed: int, max_deg: float = 5.0, max_trans: float = 0.1) -> np.ndarray:
    rs = np.random.RandomState(seed)
    c2w = np.zeros((3, 4), dtype=np.float64)
    c2w[:, :3] = _small_rotation(rs, max_deg)
    c2w[:, 3] = rs.uniform(-max_trans, max_trans, size=3)
    return c2w.astype(np.float32)


def synthetic_rays(seed: int, n_rays: int, H: int = 400, W: int = 400, focal: float = 400.0,
                   n_poses: int = 4) -> np.ndarray:
    """LLFF-shaped ray batch ``[R,3,2]`` (origin | direction in the last axis).

    Pixel coordinates U{0..W-1} x U{0..H-1}; the +0.5 half pixel and the camera
    model are those of reference utils/rays.py:25-36 (restated here in float32 numpy).
    """
    rs = np.random.RandomState(1234 + seed)
    K = synthetic_camera(H, W, focal)
    poses = np.stack([synthetic_pose(seed * 131 + p) for p in range(n_poses)])
    pid = rs.randint(0, n_poses, size=n_rays)
    px = rs.randint(0, W, size=n_rays).astype(np.float32)
    py = rs.randint(0, H, size=n_rays).astype(np.float32)
    dirs = np.stack([(px + (np.float32(0.5) - K[0, 2])) / K[0, 0],
                     -(py + (np.float32(0.5) - K[1, 2])) / K[1, 1],
                     -np.ones_like(px)], -1).astype(np.float32)
    c2w = poses[pid]
    rays_d = np.sum(dirs[:, None, :] * c2w[:, :3, :3], -1).astype(np.float32)
    rays_o = c2w[:, :3, 3].astype(np.float32)
    return np.stack([rays_o, rays_d], -1).astype(np.float32)


# ---------------------------------------------------------------------------
# the shipped 'blurfactory' configuration (BASELINE configs 2, 3, 5): PDRF coarse-to-fine levels at their real grid sizes
# (configs/evdeblurnerf_blender/tx_blurfactory_evdeblurnerf_ediprior_evcrf.txt:61-80)
# ---------------------------------------------------------------------------

BLURFACTORY_AABB = ([-1.5, -1.5, -1.0], [1.5, 1.5, 1.0])
BLURFACTORY_COARSE_VOXELS = 16777248
BLURFACTORY_FINE_VOXELS = 134217984


def blurfactory_args(N_importance: int = 64, coarse_voxels: int = BLURFACTORY_COARSE_VOXELS,
                     fine_voxels: int = BLURFACTORY_FINE_VOXELS, use_awp: bool = False):
    """The argument namespace NeRFAll reads for mode='c2f' with the config's network sizes."""
    from types import SimpleNamespace
    return SimpleNamespace(mode="c2f", multires=10, multires_views=4, use_viewdirs=True, N_importance=N_importance,
                           kernel_type="RBK", kernel_use_awp=use_awp, rgb_activate="sigmoid", sigma_activate="relu",
                           bounding_box=BLURFACTORY_AABB, coarse_num_layers=2, coarse_num_layers_color=3, coarse_hidden_dim=64,
                           coarse_hidden_dim_color=64, coarse_app_dim=32, coarse_app_n_comp=[64, 16, 16], coarse_n_voxels=coarse_voxels,
                           kernel_feat_cnl=15, fine_num_layers=2, fine_num_layers_color=3, fine_hidden_dim=256,
                           fine_hidden_dim_color=256, fine_geo_feat_dim=128, fine_app_dim=32, fine_app_n_comp=[64, 16, 16],
                           fine_n_voxels=fine_voxels)


def make_blurfactory_state_dict(seed: int = 31, coarse_voxels: int = BLURFACTORY_COARSE_VOXELS,
                                fine_voxels: int = BLURFACTORY_FINE_VOXELS, sigma_gain: float = 1.0):
    """Seed-derived parameters of both PDRF levels at the given voxel budgets (the full sizes: grids 293x293x195 and
    586x586x390, 41 M floats).  ``sigma_gain`` scales the last sigma layer so that rays are not all transparent."""
    lo, hi = BLURFACTORY_AABB
    gc, gf = pdrf_grid_size(lo, hi, coarse_voxels), pdrf_grid_size(lo, hi, fine_voxels)
    sd = OrderedDict(prefixed(make_pdrf_state_dict(seed, gc, input_ch=95, hidden_dim=64, geo_feat_dim=15), "mlp_coarse"))
    sd.update(prefixed(make_pdrf_state_dict(seed + 1, gf, input_ch=127, hidden_dim=256, geo_feat_dim=128), "mlp_fine"))
    if sigma_gain != 1.0:
        for k in list(sd):
            if k.endswith("sigma_net.1.weight"):
                sd[k] = (sd[k] * np.float32(sigma_gain)).astype(np.float32)
    return sd
