#!/usr/bin/env python
"""Full-frame test-set render (BASELINE config 5): one 400x400 LLFF view = 160 000 rays, hierarchical 64 + 128 samples,
render_kwargs_test (perturb 0, raw_noise_std 0, deterministic sample_pdf), PDRF (mode=c2f) at the blurfactory grid sizes and
the vanilla NeRF 8x256 pair.  GPU box only; with torch.distributed.run the image rows are sharded over the ranks.
    python tools/bench_fullframe.py [--precision f16] [--frames 5]"""
import argparse
import os
import sys
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from evdeblurnerf_amd import dist as D, weights as W  # noqa: E402
from evdeblurnerf_amd.renderer import NeRFAll  # noqa: E402

AABB = ([-1.5, -1.5, -1.0], [1.5, 1.5, 1.0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="f16")
    ap.add_argument("--frames", type=int, default=5)
    a = ap.parse_args()
    rank, world, local = D.init_from_env()
    torch.cuda.set_device(local)
    H = Wd = 400
    K = W.synthetic_camera()
    poses = [torch.as_tensor(W.synthetic_pose(40 + i)[:3, :4].astype(np.float32)) for i in range(a.frames)]
    kw = dict(ndc=True, near=0., far=1., use_viewdirs=True, N_samples=64, N_importance=128, perturb=0., raw_noise_std=0.)
    cv, fv = 16777248, 134217984
    gc, gf = W.pdrf_grid_size(AABB[0], AABB[1], cv), W.pdrf_grid_size(AABB[0], AABB[1], fv)
    sd = dict(W.prefixed(W.make_pdrf_state_dict(31, gc, input_ch=95, hidden_dim=64, geo_feat_dim=15), "mlp_coarse"))
    sd.update(W.prefixed(W.make_pdrf_state_dict(32, gf, input_ch=127, hidden_dim=256, geo_feat_dim=128), "mlp_fine"))
    c2f = SimpleNamespace(mode="c2f", multires=10, multires_views=4, use_viewdirs=True, N_importance=128, kernel_type="RBK",
                          kernel_use_awp=False, rgb_activate="sigmoid", sigma_activate="relu", bounding_box=AABB, coarse_num_layers=2,
                          coarse_num_layers_color=3, coarse_hidden_dim=64, coarse_hidden_dim_color=64, coarse_app_dim=32,
                          coarse_app_n_comp=[64, 16, 16], coarse_n_voxels=cv, kernel_feat_cnl=15, fine_num_layers=2, fine_num_layers_color=3,
                          fine_hidden_dim=256, fine_hidden_dim_color=256, fine_geo_feat_dim=128, fine_app_dim=32,
                          fine_app_n_comp=[64, 16, 16], fine_n_voxels=fv)
    sdn = dict(W.prefixed(W.make_nerf_state_dict(11), "mlp_coarse"))
    sdn.update(W.prefixed(W.make_nerf_state_dict(12), "mlp_fine"))
    nerf = SimpleNamespace(mode="nerf", netdepth=8, netwidth=256, multires=10, multires_views=4, use_viewdirs=True,
                           rgb_activate="sigmoid", sigma_activate="relu", N_importance=128)
    for name, args, state in (("c2f (PDRF)", c2f, sd), ("nerf 8x256", nerf, sdn)):
        model = NeRFAll(args, state, precision=a.precision).eval()
        model.render_path(H, Wd, K, 1 << 22, poses[:1], kw, shard_rows=world > 1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rgbs, depths = model.render_path(H, Wd, K, 1 << 22, poses, kw, shard_rows=world > 1)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / a.frames
        if rank == 0:
            print(f"{name:12s} [{a.precision}] {world} GPU(s): {ms:8.2f} ms per 400x400 frame (64 + 128 samples)  "
                  f"{H * Wd / ms / 1e3:6.2f} M rays/s   frame mean {float(rgbs.mean()):.4f}")
        del model


if __name__ == "__main__":
    main()
