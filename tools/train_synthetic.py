#!/usr/bin/env python
"""End-to-end training run on a synthetic scene (GPU box only): a TEACHER PDRF model (mode='c2f', random weights, seed A) renders
target colours for random LLFF-shaped rays; a STUDENT with different weights is trained on them with the library's training path
(forward under autograd, hand-written backward, Adam on networks + tri-planes, TV regulariser, device re-pack every step).
Prints the loss and the PSNR of the student against the teacher on held-out rays; exits non-zero on a non-finite value or if the
PSNR does not improve.
    python tools/train_synthetic.py [--iters 400] [--rays 2048] [--precision f16]"""
import argparse
import math
import os
import sys
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from evdeblurnerf_amd import weights as W  # noqa: E402
from evdeblurnerf_amd.renderer import NeRFAll  # noqa: E402

AABB = ([-1.5, -1.5, -1.0], [1.5, 1.5, 1.0])


def make(seed, cv, fv, precision):
    gc, gf = W.pdrf_grid_size(AABB[0], AABB[1], cv), W.pdrf_grid_size(AABB[0], AABB[1], fv)
    sd = dict(W.prefixed(W.make_pdrf_state_dict(seed, gc, input_ch=95, hidden_dim=64, geo_feat_dim=15, add_bias_color=True), "mlp_coarse"))
    sd.update(W.prefixed(W.make_pdrf_state_dict(seed + 1, gf, input_ch=127, hidden_dim=256, geo_feat_dim=128, add_bias_color=True), "mlp_fine"))
    for k in list(sd):                        # denser, more colourful fields than the default initialisation: something to learn
        if k.endswith("sigma_net.1.weight"):
            sd[k] = sd[k] * 3.0
    args = SimpleNamespace(mode="c2f", multires=10, multires_views=4, use_viewdirs=True, N_importance=64, kernel_type="RBK", kernel_use_awp=False,
                           rgb_activate="sigmoid", sigma_activate="relu", bounding_box=AABB, coarse_num_layers=2, coarse_num_layers_color=3,
                           coarse_hidden_dim=64, coarse_hidden_dim_color=64, coarse_app_dim=32, coarse_app_n_comp=[64, 16, 16], coarse_n_voxels=cv,
                           kernel_feat_cnl=15, fine_num_layers=2, fine_num_layers_color=3, fine_hidden_dim=256, fine_hidden_dim_color=256,
                           fine_geo_feat_dim=128, fine_app_dim=32, fine_app_n_comp=[64, 16, 16], fine_n_voxels=fv)
    return NeRFAll(args, sd, precision=precision), sd


def psnr(a, b):
    mse = float(((a - b) ** 2).mean())
    return -10.0 * math.log10(max(mse, 1e-12))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=400)
    ap.add_argument("--rays", type=int, default=2048)
    ap.add_argument("--precision", default="f16")
    ap.add_argument("--coarse-voxels", type=int, default=64 ** 3)
    ap.add_argument("--fine-voxels", type=int, default=128 ** 3)
    a = ap.parse_args()
    K = W.synthetic_camera()
    teacher, _ = make(101, a.coarse_voxels, a.fine_voxels, a.precision)
    teacher.eval()
    student, sd = make(202, a.coarse_voxels, a.fine_voxels, a.precision)
    student.enable_training(sd, grads_in_place=True).train()
    nets = student.get_parameters("net", not_match_re=r"basis_mat")
    grids = student.grad_vars_vol + student.get_parameters("net", match_re=r"basis_mat")
    opt = torch.optim.Adam([{"params": nets, "lr": 1e-3}, {"params": grids, "lr": 2e-2}])
    kw = dict(ndc=True, near=0., far=1., use_viewdirs=True, N_samples=64, N_importance=64, raw_noise_std=0.)
    held = torch.as_tensor(W.synthetic_rays(999, 4096), device="cuda")
    with torch.no_grad():
        held_t = teacher.render(400, 400, K, rays=held, perturb=0., **kw)[0]

    def evaluate():
        student.eval()
        with torch.no_grad():
            out = student.render(400, 400, K, rays=held, perturb=0., **kw)[0]
        student.train()
        return psnr(out, held_t)

    p0 = evaluate()
    print(f"iter    0: held-out PSNR vs teacher {p0:.2f} dB", flush=True)
    last = p0
    for it in range(1, a.iters + 1):
        rays = torch.as_tensor(W.synthetic_rays(10_000 + it, a.rays), device="cuda")
        with torch.no_grad():
            rgb_t, _, _, ex = teacher.render(400, 400, K, rays=rays, perturb=0., **kw)
        rgb, rgb0, other, _ = student(400, 400, K, 1 << 22, rays=rays, perturb=1.0, **kw)
        loss = ((rgb - rgb_t) ** 2).mean() + ((rgb0 - rgb_t) ** 2).mean() + 1e-3 * other["TV"].sum()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        lv = float(loss.detach())
        if not math.isfinite(lv):
            print(f"iter {it}: non-finite loss")
            sys.exit(1)
        if it % 50 == 0 or it == a.iters:
            last = evaluate()
            print(f"iter {it:4d}: loss {lv:.5f}  held-out PSNR vs teacher {last:.2f} dB", flush=True)
    if not last > p0 + 3.0:
        print("PSNR did not improve by 3 dB")
        sys.exit(1)
    print(f"OK: {p0:.2f} -> {last:.2f} dB in {a.iters} iterations")


if __name__ == "__main__":
    main()
