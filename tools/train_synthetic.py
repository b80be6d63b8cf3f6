#!/usr/bin/env python
"""End-to-end training run on a synthetic scene (GPU box only): a TEACHER PDRF model (mode='c2f', random weights, seed A) renders
target colours for random LLFF-shaped rays; a STUDENT with different weights is trained on them with the library's training path
(forward under autograd, hand-written backward, Adam on networks + tri-planes, TV regulariser, device re-pack every step).
Prints the loss and the PSNR of the student against the teacher on held-out rays; exits non-zero on a non-finite value or if the
PSNR does not improve.  The networks have the shipped configs' shape (rgb_add_bias off).  --precision takes a comma list: every mode is
trained from the same initial student on the same rays and the same random draws against the same (float32-grade) teacher colours, and
a table of the loss curve / held-out PSNR per mode is printed at the end (profiles/r06_convergence_by_mode.txt).
    python tools/train_synthetic.py [--iters 400] [--rays 2048] [--precision f16[,f16c,f16m,f16x3]]"""
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


def make(seed, cv, fv, precision, colourful=False):
    gc, gf = W.pdrf_grid_size(AABB[0], AABB[1], cv), W.pdrf_grid_size(AABB[0], AABB[1], fv)
    sd = dict(W.prefixed(W.make_pdrf_state_dict(seed, gc, input_ch=95, hidden_dim=64, geo_feat_dim=15, add_bias_color=False), "mlp_coarse"))
    sd.update(W.prefixed(W.make_pdrf_state_dict(seed + 1, gf, input_ch=127, hidden_dim=256, geo_feat_dim=128, add_bias_color=False), "mlp_fine"))
    for k in list(sd):                        # denser, more colourful fields than the default initialisation: something to learn
        if k.endswith("sigma_net.1.weight"):
            sd[k] = sd[k] * 3.0
        if colourful and k.endswith("color_net.2.weight"):          # the teacher: colours over the whole range (the initialisation alone is sigmoid(~0))
            sd[k] = sd[k] * 12.0
    args = SimpleNamespace(mode="c2f", multires=10, multires_views=4, use_viewdirs=True, N_importance=64, kernel_type="RBK", kernel_use_awp=False,
                           rgb_activate="sigmoid", sigma_activate="relu", bounding_box=AABB, coarse_num_layers=2, coarse_num_layers_color=3,
                           coarse_hidden_dim=64, coarse_hidden_dim_color=64, coarse_app_dim=32, coarse_app_n_comp=[64, 16, 16], coarse_n_voxels=cv,
                           kernel_feat_cnl=15, fine_num_layers=2, fine_num_layers_color=3, fine_hidden_dim=256, fine_hidden_dim_color=256,
                           fine_geo_feat_dim=128, fine_app_dim=32, fine_app_n_comp=[64, 16, 16], fine_n_voxels=fv)
    return NeRFAll(args, sd, precision=precision), sd


def psnr(a, b):
    mse = float(((a - b) ** 2).mean())
    return -10.0 * math.log10(max(mse, 1e-12))


def train_one(a, precision, teacher, K, kw, held, held_t, log_every):
    """-> (initial PSNR, [(iteration, mean loss since the last row, held-out PSNR)], seconds)"""
    import time
    torch.manual_seed(1234)                                  # the same stratification draws for every mode
    student, sd = make(202, a.coarse_voxels, a.fine_voxels, precision)
    student.enable_training(sd, grads_in_place=True).train()
    nets = student.get_parameters("net", not_match_re=r"basis_mat")
    grids = student.grad_vars_vol + student.get_parameters("net", match_re=r"basis_mat")
    opt = torch.optim.Adam([{"params": nets, "lr": 1e-3}, {"params": grids, "lr": 2e-2}])

    def evaluate():
        student.eval()
        with torch.no_grad():
            out = student.render(400, 400, K, rays=held, perturb=0., **kw)[0]
        student.train()
        return psnr(out, held_t)

    p0 = evaluate()
    print(f"[{precision}] iter    0: held-out PSNR vs teacher {p0:.2f} dB", flush=True)
    rows, acc, nacc = [], 0.0, 0
    torch.cuda.synchronize()
    t0 = time.time()
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
            print(f"[{precision}] iter {it}: non-finite loss")
            sys.exit(1)
        acc, nacc = acc + lv, nacc + 1
        if it % log_every == 0 or it == a.iters:
            rows.append((it, acc / nacc, evaluate()))
            acc, nacc = 0.0, 0
            print(f"[{precision}] iter {it:5d}: mean loss {rows[-1][1]:.4e}  held-out PSNR vs teacher {rows[-1][2]:.2f} dB", flush=True)
    torch.cuda.synchronize()
    return p0, rows, time.time() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=400)
    ap.add_argument("--rays", type=int, default=2048)
    ap.add_argument("--precision", default="f16", help="one mode or a comma list (f16,f16c,f16m,f16x3)")
    ap.add_argument("--coarse-voxels", type=int, default=64 ** 3)
    ap.add_argument("--fine-voxels", type=int, default=128 ** 3)
    ap.add_argument("--log-every", type=int, default=0, help="rows of the loss curve (default: 10 rows)")
    a = ap.parse_args()
    modes = a.precision.split(",")
    K = W.synthetic_camera()
    teacher, _ = make(101, a.coarse_voxels, a.fine_voxels, "f16x3" if len(modes) > 1 else modes[0], colourful=True)
    teacher.eval()
    kw = dict(ndc=True, near=0., far=1., use_viewdirs=True, N_samples=64, N_importance=64, raw_noise_std=0.)
    held = torch.as_tensor(W.synthetic_rays(999, 4096), device="cuda")
    with torch.no_grad():
        held_t = teacher.render(400, 400, K, rays=held, perturb=0., **kw)[0]
    log_every = a.log_every or max(a.iters // 10, 1)
    res = {m: train_one(a, m, teacher, K, kw, held, held_t, log_every) for m in modes}
    if len(modes) > 1:
        print(f"\nconvergence by training mode: {a.iters} iterations x {a.rays} rays x (64 + 64) samples, grids {a.coarse_voxels} / {a.fine_voxels} voxels, Adam 1e-3 / 2e-2, "
              "same initial student, rays, draws and float32-grade teacher for every mode")
        print("iteration | " + " | ".join(f"{m:>22s}" for m in modes))
        print("          | " + " | ".join(f"{'mean loss':>12s} {'PSNR dB':>9s}" for _ in modes))
        print(f"{0:9d} | " + " | ".join(f"{'':>12s} {res[m][0]:9.2f}" for m in modes))
        for r in range(len(res[modes[0]][1])):
            print(f"{res[modes[0]][1][r][0]:9d} | " + " | ".join(f"{res[m][1][r][1]:12.4e} {res[m][1][r][2]:9.2f}" for m in modes))
        print("seconds   | " + " | ".join(f"{res[m][2]:22.1f}" for m in modes))
        ref = res[modes[-1]][1][-1][2]
        print("final PSNR relative to the last mode listed: " + ", ".join(f"{m} {res[m][1][-1][2] - ref:+.2f} dB" for m in modes))
    for m in modes:
        p0, rows, _ = res[m]
        if not rows[-1][2] > p0 + 3.0:
            print(f"[{m}] PSNR did not improve by 3 dB")
            sys.exit(1)
    print("OK: " + ", ".join(f"{m} {res[m][0]:.2f} -> {res[m][1][-1][2]:.2f} dB" for m in modes) + f" in {a.iters} iterations")


if __name__ == "__main__":
    main()
