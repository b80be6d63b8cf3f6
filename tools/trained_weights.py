#!/usr/bin/env python
"""Trained (not seed-derived) parameters for the 8x256 NeRF of the metric workload -- GPU box only.

The parity bound of a reduced-precision mode depends on the weights: a freshly initialised network (uniform +-1/sqrt(fan_in))
shrinks its activations layer by layer and its densities stay near the bias, so rounding errors hardly reach the image.  A
trained network has densities of tens to hundreds and saturated colours.  There is no checkpoint to download, so this module
TRAINS one, with the library's own training path (forward keeping activations, hand-written backward, Adam, device re-pack),
on an analytic scene in NDC space:

    sigma(p)  = sum_j A_j exp(-|p - c_j|^2 / (2 s_j^2))          16 blobs, A_j in [30, 400], s_j in [0.03, 0.15]
    colour(p) = 0.5 + 0.5 sin(F_j p + phi_j)  blended by the blobs' densities, tinted by the view direction

rendered by the reference's compositing rule (nerf.py:74-129, through evd_raw2outputs) on the same stratified samples.
`train_nerf(...)` returns the trained state dict (reference key names, float32 numpy) and a small report; tests and
bench.py then compare every arithmetic mode of the render kernels with the CPU oracle ON THESE WEIGHTS.

    python tools/trained_weights.py [--iters 600]       # prints the report
"""
from __future__ import annotations

import argparse
import os
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from evdeblurnerf_amd import weights as W  # noqa: E402

NERF_ARGS = dict(mode="nerf", netdepth=8, netwidth=256, multires=10, multires_views=4, use_viewdirs=True,
                 rgb_activate="sigmoid", sigma_activate="relu")


def scene(seed=7, n_blobs=16, device="cuda", smin=0.03, smax=0.15):
    rs = np.random.RandomState(seed)
    t = lambda a: torch.as_tensor(np.asarray(a, np.float32), device=device)
    return {"c": t(rs.uniform([-0.8, -0.8, -0.6], [0.8, 0.8, 0.9], (n_blobs, 3))), "s": t(rs.uniform(smin, smax, n_blobs)),
            "A": t(rs.uniform(30.0, 400.0, n_blobs)), "F": t(rs.uniform(-6.0, 6.0, (n_blobs, 3, 3))),
            "phi": t(rs.uniform(0, 6.28, (n_blobs, 3))), "tint": t(rs.uniform(-0.6, 0.6, (3, 3)))}


def scene_raw(sc, pts, viewdirs):
    """raw [R,S,4] = (colour logits, density) of the analytic field at pts [R,S,3] for unit view directions [R,3]"""
    d2 = ((pts[:, :, None, :] - sc["c"]) ** 2).sum(-1)                                  # [R,S,B]
    dens = sc["A"] * torch.exp(-0.5 * d2 / sc["s"] ** 2)
    col = 0.5 + 0.5 * torch.sin(torch.einsum("bij,rsj->rsbi", sc["F"], pts) + sc["phi"])  # [R,S,B,3]
    wsum = dens.sum(-1, keepdim=True) + 1e-3
    col = (col * dens[..., None]).sum(2) / wsum + 0.15 * torch.tanh(viewdirs @ sc["tint"])[:, None, :]
    col = col.clamp(0.02, 0.98)
    return torch.cat([torch.log(col / (1 - col)), dens.sum(-1, keepdim=True)], -1).contiguous()


def train_nerf(iters=1000, rays_per_iter=4096, samples=128, seed=21, precision="f16", lr=5e-4, verbose=False):
    """-> (state dict with 'mlp_coarse.' keys, report dict).  Starts from the seed-derived parameters the bench uses."""
    from evdeblurnerf_amd.renderer import NeRFAll
    dev = "cuda"
    sd0 = W.prefixed(W.make_nerf_state_dict(seed), "mlp_coarse")
    args = SimpleNamespace(N_importance=0, **NERF_ARGS)
    model = NeRFAll(args, sd0, precision=precision).train()
    flat, _ = model.trainable_parameters(sd0)
    opt = torch.optim.Adam([flat], lr=lr)
    K = W.synthetic_camera()
    sc = scene(device=dev)
    net = model.mlp_coarse
    losses = []
    for it in range(iters):
        rays = torch.as_tensor(W.synthetic_rays(50_000 + it, rays_per_iter), device=dev)
        with torch.no_grad():
            rb = NeRFAll.ray_batch_train(400, 400, K, rays)
            t_rand = torch.rand((rays_per_iter, samples), device=dev)
        out = model.render_rays_train(rb, flat, None, samples, 0, perturb=1.0, t_rand=t_rand)
        with torch.no_grad():
            z = out["z_vals"]
            pts = rb[:, None, 0:3] + rb[:, None, 3:6] * z[..., None]
            target = net.raw2outputs(scene_raw(sc, pts, rb[:, 8:11]), z, rb[:, 3:6].contiguous())[0]
        loss = ((out["rgb_map"] - target) ** 2).mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
        if verbose and (it % 100 == 0 or it == iters - 1):
            print(f"iter {it:4d}: mse {losses[-1]:.5f}", flush=True)
    model.eval()
    sd = {"mlp_coarse." + k: v.detach().cpu().numpy().astype(np.float32).copy() for k, v in net.unflatten(flat).items()}
    hid = [np.abs(sd[f"mlp_coarse.pts_linears.{i}.weight"]).mean() / np.abs(sd0[f"mlp_coarse.pts_linears.{i}.weight"]).mean() for i in range(8)]
    with torch.no_grad():
        rays = torch.as_tensor(W.synthetic_rays(77, 1024), device=dev)
        rb = NeRFAll.ray_batch_train(400, 400, K, rays)
        z = torch.linspace(0, 1, samples, device=dev).expand(1024, samples).contiguous()
        raw, _ = net.mlpforward(rb, z, precision="f32")
    report = {"iters": iters, "mse_first": float(np.mean(losses[:5])), "mse_last": float(np.mean(losses[-20:])),
              "sigma_max": float(raw[..., 3].max()), "sigma_p99": float(torch.quantile(raw[..., 3].flatten().clamp(min=0), 0.99)),
              "hidden_weight_growth": [float(h) for h in hid], "precision": precision}
    return sd, report


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=1000)
    ap.add_argument("--save", default=None, help="write the trained state dict (.npz)")
    ap.add_argument("--check", action="store_true", help="RGB L-inf of every mode vs the CPU oracle at 4096 x 128 on the trained weights")
    a = ap.parse_args()
    t0 = time.time()
    sd, rep = train_nerf(a.iters, verbose=True)
    rep["train_s"] = time.time() - t0
    print(rep)
    if a.save:
        np.savez(a.save, **sd)
    if a.check:
        from oracle import oracle as O
        from evdeblurnerf_amd.renderer import NeRFAll
        rays = W.synthetic_rays(100, 4096)
        ref = O.render_nerf(O.Nerf(sd, "mlp_coarse."), None, O.make_cfg(N_samples=128), rays)["rgb"]
        args = SimpleNamespace(N_importance=0, **NERF_ARGS)
        for p in ("f32", "f16x3", "f16c", "f16", "bf16"):
            out = NeRFAll(args, sd, precision=p).eval().render(400, 400, W.synthetic_camera(), rays=torch.as_tensor(rays, device="cuda"), N_samples=128,
                                                               ndc=True, near=0., far=1., use_viewdirs=True, N_importance=0, retraw=False)[0]
            print(f"trained {a.iters}: {p:6s} RGB L-inf vs oracle {float(np.abs(out.cpu().numpy() - ref).max()):.3e}", flush=True)
