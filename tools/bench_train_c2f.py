#!/usr/bin/env python
"""Timing of one mode='c2f' training iteration (forward under autograd + backward, no optimizer) at the blurfactory
configuration's sizes: coarse / fine grids of ~16.8 M / ~134 M voxels, n_comp (64,16,16), 64 coarse + 64 importance samples.
GPU box only.
    python tools/bench_train_c2f.py [--rays 4096] [--iters 10] [--precision f16]"""
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
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--precision", default="f16")
    ap.add_argument("--coarse-voxels", type=int, default=16777248)
    ap.add_argument("--fine-voxels", type=int, default=134217984)
    ap.add_argument("--n-importance", type=int, default=64)
    a = ap.parse_args()
    gc = W.pdrf_grid_size(AABB[0], AABB[1], a.coarse_voxels)
    gf = W.pdrf_grid_size(AABB[0], AABB[1], a.fine_voxels)
    sd = dict(W.prefixed(W.make_pdrf_state_dict(31, gc, input_ch=95, hidden_dim=64, geo_feat_dim=15), "mlp_coarse"))
    sd.update(W.prefixed(W.make_pdrf_state_dict(32, gf, input_ch=127, hidden_dim=256, geo_feat_dim=128), "mlp_fine"))
    args = SimpleNamespace(mode="c2f", multires=10, multires_views=4, use_viewdirs=True, N_importance=a.n_importance,
                           kernel_type="RBK", kernel_use_awp=False, rgb_activate="sigmoid", sigma_activate="relu",
                           bounding_box=AABB, coarse_num_layers=2, coarse_num_layers_color=3, coarse_hidden_dim=64,
                           coarse_hidden_dim_color=64, coarse_app_dim=32, coarse_app_n_comp=[64, 16, 16], coarse_n_voxels=a.coarse_voxels,
                           kernel_feat_cnl=15, fine_num_layers=2, fine_num_layers_color=3, fine_hidden_dim=256,
                           fine_hidden_dim_color=256, fine_geo_feat_dim=128, fine_app_dim=32, fine_app_n_comp=[64, 16, 16],
                           fine_n_voxels=a.fine_voxels)
    model = NeRFAll(args, sd, precision=a.precision).train()
    pc, pf = model.trainable_parameters(sd)
    R = a.rays
    rs = np.random.RandomState(0)
    rb = np.zeros((R, 11), np.float32)
    rb[:, 0:3] = rs.uniform(-0.3, 0.3, (R, 3)) + np.array([0, 0, 0.9])
    d = rs.normal(size=(R, 3)) * 0.35 + np.array([0, 0, -1.0])
    rb[:, 3:6] = d
    rb[:, 6], rb[:, 7] = 0.1, 1.7
    rb[:, 8:11] = d / np.linalg.norm(d, axis=-1, keepdims=True)
    rb = torch.as_tensor(rb, device="cuda")
    target = torch.rand((R, 3), device="cuda")
    params = [pc["net"], pf["net"]] + pc["grids"] + pf["grids"]

    def step(backward=True):
        out = model.render_rays_train(rb, pc, pf, 64, a.n_importance, perturb=1.0)
        loss = ((out["rgb_map"] - target) ** 2).mean() + ((out["rgb0"] - target) ** 2).mean() + 0.01 * model.tv_loss_train(pc, pf)
        if backward:
            for p in params:
                p.grad = None
            loss.backward()
        return loss

    for mode, bw in (("forward under autograd", False), ("forward + backward", True)):
        for _ in range(2):
            step(bw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(a.iters):
            step(bw)
        e1.record()
        e1.synchronize()
        print(f"c2f training iteration, {mode}, {a.precision}: {R} rays x (64 + {a.n_importance}) samples: {e0.elapsed_time(e1) / a.iters:.3f} ms")


if __name__ == "__main__":
    main()
