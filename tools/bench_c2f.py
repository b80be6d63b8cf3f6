#!/usr/bin/env python
"""Timing of the PDRF (mode=c2f) render at the blurfactory configuration's sizes (BASELINE configs 2/3): coarse grid
n_voxels 16 777 248-ish and fine 134 217 984-ish voxels over aabb (+-1.5, +-1.5, +-1.0), n_comp (64,16,16), 64 coarse +
64 importance samples (128 at the fine level).  GPU box only.
    python tools/bench_c2f.py [--rays 4096] [--iters 20] [--precision f16]"""
import argparse
import os
import sys
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from evdeblurnerf_amd import weights as W  # noqa: E402
from evdeblurnerf_amd.renderer import NeRFAll  # noqa: E402

AABB = ([-1.5, -1.5, -1.0], [1.5, 1.5, 1.0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--precision", default="f16")
    ap.add_argument("--coarse-voxels", type=int, default=16777248)
    ap.add_argument("--fine-voxels", type=int, default=134217984)
    ap.add_argument("--n-importance", type=int, default=64)
    a = ap.parse_args()
    gc = W.pdrf_grid_size(AABB[0], AABB[1], a.coarse_voxels)
    gf = W.pdrf_grid_size(AABB[0], AABB[1], a.fine_voxels)
    print("grids", gc, gf, flush=True)
    sd = dict(W.prefixed(W.make_pdrf_state_dict(31, gc, input_ch=95, hidden_dim=64, geo_feat_dim=15), "mlp_coarse"))
    sd.update(W.prefixed(W.make_pdrf_state_dict(32, gf, input_ch=127, hidden_dim=256, geo_feat_dim=128), "mlp_fine"))
    args = SimpleNamespace(mode="c2f", multires=10, multires_views=4, use_viewdirs=True, N_importance=a.n_importance,
                           kernel_type="RBK", kernel_use_awp=False, rgb_activate="sigmoid", sigma_activate="relu",
                           bounding_box=AABB, coarse_num_layers=2, coarse_num_layers_color=3, coarse_hidden_dim=64,
                           coarse_hidden_dim_color=64, coarse_app_dim=32, coarse_app_n_comp=[64, 16, 16], coarse_n_voxels=a.coarse_voxels,
                           kernel_feat_cnl=15, fine_num_layers=2, fine_num_layers_color=3, fine_hidden_dim=256,
                           fine_hidden_dim_color=256, fine_geo_feat_dim=128, fine_app_dim=32, fine_app_n_comp=[64, 16, 16],
                           fine_n_voxels=a.fine_voxels)
    model = NeRFAll(args, sd, precision=a.precision).eval()
    rays = torch.as_tensor(W.synthetic_rays(5, a.rays), device="cuda")
    K = W.synthetic_camera()
    kw = dict(ndc=True, near=0., far=1., use_viewdirs=True, N_samples=64, N_importance=a.n_importance, retraw=False, perturb=0., raw_noise_std=0.)
    for _ in range(3):
        model.render(400, 400, K, rays=rays, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(a.iters):
        model.render(400, 400, K, rays=rays, **kw)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    print(f"c2f render {a.precision}: {a.rays} rays x (64 + {a.n_importance}) samples: {ms:.3f} ms/step  {a.rays / ms / 1e3:.3f} M rays/s")


if __name__ == "__main__":
    main()
