#!/usr/bin/env python
"""Trained (not seed-derived) parameters for the shipped PDRF configuration (mode='c2f', blurfactory grid sizes) -- GPU box only.

Same purpose as tools/trained_weights.py has for the 8x256 NeRF: the parity bound of a reduced-precision mode depends on the
weights, and there is no checkpoint to download, so this module TRAINS both PDRF levels (tri-planes 293x293x195 / 586x586x390 +
their sigma / colour networks, 42 M parameters) with the library's own training path -- NeRFAll.forward's training branch under
autograd, hand-written backward kernels, Adam on networks and grids, TV regulariser, device re-pack every step -- on the
analytic scene of trained_weights.scene(): per-ray target colours are the scene composited by the reference's rule
(voxnerf.py:167-201: last alpha forced to 1) on a dense ladder of 192 samples along the NDC ray.

    train_c2f(iters)            -> (state dict under the reference's keys, float32 numpy; report dict)
    c2f_parity(O, sd, ...)      -> RGB L-inf of each arithmetic mode vs oracle.render_c2f on a slice of the rays
    python tools/trained_c2f.py [--iters 3000]          # trains, prints the report and the parity table
"""
from __future__ import annotations

import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from evdeblurnerf_amd import weights as W  # noqa: E402
from trained_weights import scene  # noqa: E402

KW = dict(ndc=True, near=0., far=1., use_viewdirs=True, N_samples=64, N_importance=64, raw_noise_std=0.)


def scene_rgb(sc, rb, n_dense=192):
    """per-ray colour [R,3] of the analytic field: dense composite along the packed ray batch rb [R,11] (NDC o, d, near, far, viewdirs)"""
    R = rb.shape[0]
    z = torch.linspace(0., 1., n_dense, device=rb.device).expand(R, n_dense)
    pts = rb[:, None, 0:3] + rb[:, None, 3:6] * z[..., None]
    d2 = ((pts[:, :, None, :] - sc["c"]) ** 2).sum(-1)                                     # [R,S,B]
    dens = sc["A"] * torch.exp(-0.5 * d2 / sc["s"] ** 2)
    col = 0.5 + 0.5 * torch.sin(torch.einsum("bij,rsj->rsbi", sc["F"], pts) + sc["phi"])   # [R,S,B,3]
    col = (col * dens[..., None]).sum(2) / (dens.sum(-1, keepdim=True) + 1e-3) + 0.15 * torch.tanh(rb[:, 8:11] @ sc["tint"])[:, None, :]
    col = col.clamp(0.02, 0.98)
    sigma = dens.sum(-1)
    dist = (z[:, 1:] - z[:, :-1]) * torch.linalg.norm(rb[:, 3:6], dim=-1, keepdim=True)
    alpha = torch.cat([1. - torch.exp(-sigma[:, :-1] * dist), torch.ones((R, 1), device=rb.device)], -1)
    T = torch.cumprod(torch.cat([torch.ones((R, 1), device=rb.device), 1. - alpha[:, :-1]], -1), -1)
    return ((alpha * T)[..., None] * col).sum(1)


def train_c2f(iters=3000, rays_per_iter=4096, seed=31, precision="f16", lr_net=1e-3, lr_grid=2e-2, verbose=False):
    """-> (state dict, report).  Starts from the seed-derived blurfactory parameters the tests and the bench use."""
    from evdeblurnerf_amd.renderer import NeRFAll
    dev = "cuda"
    sd0 = W.make_blurfactory_state_dict(seed, sigma_gain=3.0)
    model = NeRFAll(W.blurfactory_args(64), sd0, precision=precision)
    model.enable_training(sd0, grads_in_place=True).train()
    nets = model.get_parameters("net", not_match_re=r"basis_mat")
    grids = model.grad_vars_vol + model.get_parameters("net", match_re=r"basis_mat")
    opt = torch.optim.Adam([{"params": nets, "lr": lr_net}, {"params": grids, "lr": lr_grid}])
    K = W.synthetic_camera()
    sc = scene(device=dev)
    losses = []
    torch.cuda.synchronize()
    t0 = time.time()
    for it in range(iters):
        rays = torch.as_tensor(W.synthetic_rays(70_000 + it, rays_per_iter), device=dev)
        with torch.no_grad():
            target = scene_rgb(sc, NeRFAll.ray_batch_train(400, 400, K, rays))
        rgb, rgb0, other, _ = model(400, 400, K, 1 << 22, rays=rays, perturb=1.0, **KW)
        loss = ((rgb - target) ** 2).mean() + ((rgb0 - target) ** 2).mean() + 1e-3 * other["TV"].sum()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        if it % 50 == 0 or it == iters - 1:
            losses.append(float(loss.detach()))
            if verbose and (it % 500 == 0 or it == iters - 1):
                print(f"iter {it:5d}: loss {losses[-1]:.5f}", flush=True)
    torch.cuda.synchronize()
    dt = time.time() - t0
    model.eval()
    sd = {k: v.detach().cpu().numpy().astype(np.float32).copy() for k, v in model.state_dict().items()}
    growth = {k: float(np.abs(sd[k]).mean() / max(np.abs(np.asarray(sd0[k])).mean(), 1e-30))
              for k in ("mlp_fine.sigma_net.0.weight", "mlp_fine.sigma_net.1.weight", "mlp_fine.color_net.1.weight", "mlp_fine.app_plane.0")}
    report = {"iters": iters, "loss_first": losses[0], "loss_last": float(np.mean(losses[-3:])), "train_s": dt, "ms_per_iter": 1e3 * dt / max(iters, 1),
              "growth": growth, "precision": precision}
    del model
    return sd, report


def c2f_parity(O, sd, precisions=("f32", "f16x3", "f16", "bf16"), n_rays=4096, n_oracle=256, Ni=64, rays=None, env=None):
    """RGB L-inf (fine, coarse) of each mode vs the oracle on n_oracle rays spread over the batch.  env: {precision: {VAR: value}} set while
    that mode's model is built and rendered (developer switches such as EVD_F32_GRIDS)."""
    from evdeblurnerf_amd.renderer import NeRFAll
    lo, hi = W.BLURFACTORY_AABB
    gc, gf = W.pdrf_grid_size(lo, hi, W.BLURFACTORY_COARSE_VOXELS), W.pdrf_grid_size(lo, hi, W.BLURFACTORY_FINE_VOXELS)
    vc = O.Voxel(sd, "mlp_coarse.", gc, lo + hi, input_ch=95)
    vf = O.Voxel(sd, "mlp_fine.", gf, lo + hi, input_ch=127, hidden_dim=256, geo_feat_dim=128, rgb_act="none")
    K = W.synthetic_camera()
    rays = W.synthetic_rays(5, n_rays) if rays is None else rays
    R = rays.shape[0]
    idx = np.linspace(0, R - 1, n_oracle).astype(np.int64)
    ref = O.render_c2f(vc, vf, O.make_cfg(N_samples=64, N_importance=Ni), rays[idx])
    kw = dict(KW, N_importance=Ni, retraw=True, perturb=0.)
    out = {}
    for p in precisions:
        prec = p.split("+")[0]
        model = NeRFAll(W.blurfactory_args(Ni), sd, precision=prec).eval()
        rgb, _, _, ex = model.render(400, 400, K, rays=torch.as_tensor(rays, device="cuda"), **kw)
        err = np.abs(rgb.cpu().numpy()[idx] - ref["rgb"]).max(-1)
        e0 = float(np.abs(ex["rgb0"].cpu().numpy()[idx] - ref["rgb0"]).max())
        out[p] = {"fine": float(err.max()), "coarse": e0}
        # Importance samples are an ill-conditioned function of the coarse weights (tests/conftest.py z_mismatch: a 1e-7 change of a weight can move a
        # sample of a nearly empty bin by a bin): a ray whose merged sample positions differ from the oracle's renders a DIFFERENT quadrature of the
        # same field, in every arithmetic mode alike -- reported apart from the error on the rays that carry the oracle's samples
        if "z_vals" in ex:
            moved = np.abs(ex["z_vals"].cpu().numpy()[idx].astype(np.float64) - ref["z_vals"]).max(-1) > 5e-6
            out[p]["rays_with_moved_importance_samples"] = int(moved.sum())
            out[p]["fine_on_the_oracles_samples"] = float(err[~moved].max()) if (~moved).any() else 0.0
        del model
    return out, {"rgb_std": float(np.std(ref["rgb"])), "sigma_note": "oracle slice of %d rays" % n_oracle}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=3000)
    ap.add_argument("--precision", default="f16")
    ap.add_argument("--save", default=None, help="write the trained state dict (.npz, ~165 MB)")
    ap.add_argument("--load", default=None, help="skip the training, read a state dict written by --save")
    ap.add_argument("--modes", default="f32,f16x3,f16,bf16")
    a = ap.parse_args()
    from oracle import oracle as O
    modes = tuple(a.modes.split(","))
    if a.load:
        sd = dict(np.load(a.load))
    else:
        sd, rep = train_c2f(a.iters, precision=a.precision, verbose=True)
        print("report:", rep, flush=True)
        if a.save:
            np.savez(a.save, **sd)
        seed_sd = W.make_blurfactory_state_dict(31, sigma_gain=3.0)
        print("seed   :", c2f_parity(O, seed_sd, modes), flush=True)
    print("trained:", c2f_parity(O, sd, modes), flush=True)
