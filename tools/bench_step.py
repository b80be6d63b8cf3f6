#!/usr/bin/env python
"""Forward pass of one blurfactory training iteration (BASELINE configs 2/3; reference call stack SURVEY 3.1,
run_nerf.py:438-591), synthetic inputs, GPU box only:
    blur batch   1024 pixels x P = 10 sub-exposure rays -> c2f render (64 + 64) -> fused blur-loss reduction
                 (rbk weighted sums with the RBK and AWP weights, gamma CRF, image / pts0-EDI-prior terms)
    event batch  2 x 4096 rays (start, end) -> c2f render -> fused event-loss reduction (learnable event-CRF with the
                 pos/neg polarity features, rec601 luma, EGM log-difference loss)
    TV regulariser over all planes / lines of both levels
    one all-reduce-ready packed partial vector per loss (evdeblurnerf_amd/dist.py)
The blur-kernel network (RigidBlurringModel) and AWP stay PyTorch in the reference's caller and are replaced here by
synthetic warped rays / weights of their output shapes.
    python tools/bench_step.py [--precision f16] [--iters 20]"""
import argparse
import os
import sys
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from evdeblurnerf_amd import weights as W  # noqa: E402
from evdeblurnerf_amd.losses import (blur_loss_from_partials, blur_loss_partials, event_loss_from_partials,  # noqa: E402
                                     event_loss_partials)
from evdeblurnerf_amd.renderer import NeRFAll  # noqa: E402
from evdeblurnerf_amd.tonemapping import CRF  # noqa: E402

AABB = ([-1.5, -1.5, -1.0], [1.5, 1.5, 1.0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="f16")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--pixels", type=int, default=1024)
    ap.add_argument("--events", type=int, default=4096)
    ap.add_argument("--P", type=int, default=10)
    a = ap.parse_args()
    cv, fv = 16777248, 134217984
    gc, gf = W.pdrf_grid_size(AABB[0], AABB[1], cv), W.pdrf_grid_size(AABB[0], AABB[1], fv)
    sd = dict(W.prefixed(W.make_pdrf_state_dict(31, gc, input_ch=95, hidden_dim=64, geo_feat_dim=15), "mlp_coarse"))
    sd.update(W.prefixed(W.make_pdrf_state_dict(32, gf, input_ch=127, hidden_dim=256, geo_feat_dim=128), "mlp_fine"))
    args = SimpleNamespace(mode="c2f", multires=10, multires_views=4, use_viewdirs=True, N_importance=64,
                           kernel_type="RBK", kernel_use_awp=False, rgb_activate="sigmoid", sigma_activate="relu",
                           bounding_box=AABB, coarse_num_layers=2, coarse_num_layers_color=3, coarse_hidden_dim=64,
                           coarse_hidden_dim_color=64, coarse_app_dim=32, coarse_app_n_comp=[64, 16, 16], coarse_n_voxels=cv,
                           kernel_feat_cnl=15, fine_num_layers=2, fine_num_layers_color=3, fine_hidden_dim=256,
                           fine_hidden_dim_color=256, fine_geo_feat_dim=128, fine_app_dim=32, fine_app_n_comp=[64, 16, 16],
                           fine_n_voxels=fv)
    model = NeRFAll(args, sd, precision=a.precision).train(True)
    crf_rgb = CRF("gamma")
    crf_ev = CRF("learn", state_dict=W.make_crf_state_dict(5, extra_features=2), extra_features=2)
    dev = "cuda"
    K = W.synthetic_camera()
    R, P, E = a.pixels, a.P, a.events
    rs = np.random.RandomState(0)
    blur_rays = torch.as_tensor(W.synthetic_rays(1, R * P), device=dev)            # kernelsnet output shape [R*P,3,2]
    ev_start = torch.as_tensor(W.synthetic_rays(2, E), device=dev)
    ev_end = torch.as_tensor(W.synthetic_rays(3, E), device=dev)
    w1 = torch.softmax(torch.randn((R, P), device=dev), -1)
    w2 = torch.softmax(torch.randn((R, P), device=dev), -1)
    tgt, tgt0 = torch.rand((R, 3), device=dev), torch.rand((R, 3), device=dev)
    cum_neg = -torch.randint(0, 4, (E,), device=dev).float()
    cum_pos = torch.randint(0, 4, (E,), device=dev).float()
    kw = dict(ndc=True, near=0., far=1., use_viewdirs=True, N_samples=64, N_importance=64, retraw=True, perturb=1.0, raw_noise_std=0.)

    def step():
        rgb, depth, acc, ex = model.render(400, 400, K, rays=blur_rays, **kw)
        pb, _ = blur_loss_partials(crf_rgb, rgb.reshape(R, P, 3), w1, tgt, rgb0_p=ex["rgb0"].reshape(R, P, 3), w2=w2, target_pts0=tgt0)
        s1, _, _, e1 = model.render(400, 400, K, rays=ev_start, **kw)
        s2, _, _, e2 = model.render(400, 400, K, rays=ev_end, **kw)
        pe = event_loss_partials(crf_ev, s1, s2, cum_neg, cum_pos, 0.2, 0.2, start0=e1["rgb0"], end0=e2["rgb0"], add_bii="pos-neg")
        tv = model.mlp_coarse.TV_loss_app() + model.mlp_fine.TV_loss_app()
        loss, _ = blur_loss_from_partials(pb, fine_loss_weight=0.5, w_pts0=0.1)
        return loss + 0.1 * event_loss_from_partials(pe) + 5.0 * tv

    for _ in range(3):
        l = step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(a.iters):
        l = step()
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    nr = R * P + 2 * E
    print(f"blurfactory iteration forward [{a.precision}]: {nr} rays x (64 + 64) samples + losses + TV: {ms:.3f} ms  "
          f"({nr / ms / 1e3:.2f} M rays/s); loss = {float(l):.5f}")


if __name__ == "__main__":
    main()
