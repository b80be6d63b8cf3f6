#!/usr/bin/env python
"""One WHOLE blurfactory training iteration (reference call stack SURVEY 3.1, run_nerf.py:423-601) on synthetic inputs, GPU only:
    blur batch   1024 pixels -> rigid blur kernel (a small PyTorch module standing in for RigidBlurringModel: P = 10 warped rays
                 per pixel + composition weights, learnable) -> c2f render under autograd -> fused blur-loss reduction
    event batch  2 x 4096 rays (start, end) -> c2f render under autograd -> fused event-loss reduction with the learnable event-CRF
    TV regulariser; backward through everything (hand-written kernels); Adam on both levels' networks and grids, the kernel and
    the CRF; parameters pushed back into the library
    python tools/bench_train_step.py [--precision f16] [--iters 10]"""
import argparse
import os
import sys
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from evdeblurnerf_amd import weights as W  # noqa: E402
from evdeblurnerf_amd.losses import (blur_loss_from_partials, blur_loss_partials_autograd, event_loss_from_partials,  # noqa: E402
                                     event_loss_partials_autograd)
from evdeblurnerf_amd.renderer import NeRFAll  # noqa: E402
from evdeblurnerf_amd.tonemapping import CRF  # noqa: E402

AABB = ([-1.5, -1.5, -1.0], [1.5, 1.5, 1.0])


class RigidKernel(torch.nn.Module):
    def __init__(self, P):
        super().__init__()
        g = torch.Generator().manual_seed(0)
        self.P = P
        self.trans = torch.nn.Parameter(0.01 * torch.randn(P, 3, generator=g))
        self.rot = torch.nn.Parameter(0.01 * torch.randn(P, 3, generator=g))
        self.logit = torch.nn.Parameter(torch.zeros(P))

    def forward(self, H, W_, K, rays, rays_info, feats=None, return_img_embed=False):
        o, d = rays[..., 0], rays[..., 1]
        o2 = o[:, None] + self.trans[None]
        d2 = d[:, None] + torch.cross(self.rot[None].expand(d.shape[0], -1, -1), d[:, None].expand(-1, self.P, -1), dim=-1)
        extra = {"img_embed": torch.ones((o.shape[0], 4), device=o.device)} if return_img_embed else {}
        return torch.stack([o2, d2], -1), torch.softmax(self.logit, 0)[None].expand(o.shape[0], -1), None, extra


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="f16")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--pixels", type=int, default=1024)
    ap.add_argument("--events", type=int, default=4096)
    ap.add_argument("--P", type=int, default=10)
    ap.add_argument("--awp", choices=["none", "fused", "torch"], default="none",
                    help="the shipped configs' adaptive weight proposal on the blur batch (kernel_use_awp): fused = evdeblurnerf_amd.awp.FusedAWP around a "
                         "module with the reference's surface (tools/awp_standin.py), torch = that module's plain PyTorch forward on depth_feature")
    ap.add_argument("--graph", action="store_true", help="--awp fused: the per-ray remainder of the AWP module as a captured hipGraph each way (measured slower than eager)")
    ap.add_argument("--no-graph", action="store_true", help="(the default now; kept for old command lines)")
    ap.add_argument("--plain-autograd", action="store_true",
                    help="parameter / grid gradients returned to autograd (the library's default) instead of accumulated in place by the backward kernels "
                         "(NeRFAll.enable_training(grads_in_place=True): what a run_nerf.py-style loop opts into)")
    ap.add_argument("--merge-events", action="store_true", help="render the event batch's start and end rays in one call (integration option; the reference calls nerf() twice)")
    ap.add_argument("--mam", choices=["mean", "corr"], default="corr",
                    help="the AWP module's motion aggregation: corr = the reference's MotionAggregationModule structure (MAMLike), mean = a small stand-in")
    a = ap.parse_args()
    ms, nr, loss = run(a)
    print(f"blurfactory TRAINING iteration [{a.precision}]: {nr} rays x (64 + 64) samples, losses, TV, backward, Adam, re-pack: {ms:.2f} ms "
          f"({nr / ms / 1e3:.2f} M rays/s); loss = {loss:.5f}")


def run(a):
    """a: namespace with precision, iters, pixels, events, P -> (ms per iteration, rays per iteration, last loss)"""
    cv, fv = 16777248, 134217984
    gc, gf = W.pdrf_grid_size(AABB[0], AABB[1], cv), W.pdrf_grid_size(AABB[0], AABB[1], fv)
    sd = dict(W.prefixed(W.make_pdrf_state_dict(31, gc, input_ch=95, hidden_dim=64, geo_feat_dim=15), "mlp_coarse"))
    sd.update(W.prefixed(W.make_pdrf_state_dict(32, gf, input_ch=127, hidden_dim=256, geo_feat_dim=128), "mlp_fine"))
    args = SimpleNamespace(mode="c2f", multires=10, multires_views=4, use_viewdirs=True, N_importance=64, kernel_type="RBK", kernel_use_awp=False,
                           rgb_activate="sigmoid", sigma_activate="relu", bounding_box=AABB, coarse_num_layers=2, coarse_num_layers_color=3,
                           coarse_hidden_dim=64, coarse_hidden_dim_color=64, coarse_app_dim=32, coarse_app_n_comp=[64, 16, 16], coarse_n_voxels=cv,
                           kernel_feat_cnl=15, fine_num_layers=2, fine_num_layers_color=3, fine_hidden_dim=256, fine_hidden_dim_color=256,
                           fine_geo_feat_dim=128, fine_app_dim=32, fine_app_n_comp=[64, 16, 16], fine_n_voxels=fv)
    dev = "cuda"
    kern = RigidKernel(a.P).to(dev)
    awp_mode = getattr(a, "awp", "none")
    awpnet = None
    if awp_mode != "none":
        from awp_standin import RefLikeAWP
        from evdeblurnerf_amd.awp import FusedAWP
        awpnet = RefLikeAWP(P=a.P, view_ch=4, mam=getattr(a, "mam", "corr")).to(dev)
        if awp_mode == "fused":
            awpnet = FusedAWP(awpnet, precision=a.precision if a.precision in ("f16", "bf16") else "f16", graph_per_ray=bool(getattr(a, "graph", False)))
    model = NeRFAll(args, sd, kernelsnet=kern, awpnet=awpnet, precision=a.precision).enable_training(sd, grads_in_place=not getattr(a, "plain_autograd", False)).train()
    model.use_awp = awpnet is not None
    crf_rgb = CRF("gamma")
    crf_ev = CRF("learn", state_dict=W.make_crf_state_dict(5, extra_features=2), extra_features=2)
    crf_flat = crf_ev.flat_params(dev)
    # run_nerf.py:245-263 builds torch.optim.Adam over the same parameter groups; fused=True is that optimizer's single-kernel implementation
    opt = torch.optim.Adam([{"params": model.parameters(), "lr": 5e-4}, {"params": [crf_flat], "lr": 1e-4}], fused=not bool(os.environ.get("EVD_ADAM_FOREACH")))
    K = W.synthetic_camera()
    R, E = a.pixels, a.events
    # data-parallel mode (bench.py --gpus N, weak scaling: every rank trains on its OWN batch of this size): the packed loss partials are
    # all-reduced through autograd (the path's one exchange, dist.all_reduce_partials), the parameter / grid gradients by dist.GradReducer --
    # on the persistent flat gradient buffers themselves in the in-place mode (no packing copies around the 150 MB of grid gradients)
    ddp = bool(getattr(a, "dist", False))
    rank = 0
    red = None
    if ddp:
        import torch.distributed as tdist
        from evdeblurnerf_amd import dist as D
        rank = tdist.get_rank()
        # .attach: a level's buffers are all-reduced as soon as its last backward node of the iteration has run (behind them: the coarse
        # level's last scatter / networks, the blur kernel's backward); what is still waited for after loss.backward() is the exposed part
        red = D.GradReducer(list(model.parameters()) + [crf_flat], flat_buffers=model.grad_buffers())
        if not os.environ.get("EVD_NO_EARLY_ALLREDUCE"):
            red.attach(model)
    blur_rays = torch.as_tensor(W.synthetic_rays(1 + 10 * rank, R), device=dev)
    ev_start = torch.as_tensor(W.synthetic_rays(2 + 10 * rank, E), device=dev)
    ev_end = torch.as_tensor(W.synthetic_rays(3 + 10 * rank, E), device=dev)
    ar_ev = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)] if ddp else None
    ar_ms = []
    tgt, tgt0 = torch.rand((R, 3), device=dev), torch.rand((R, 3), device=dev)
    cum_neg = -torch.randint(0, 4, (E,), device=dev).float()
    cum_pos = torch.randint(0, 4, (E,), device=dev).float()
    kw = dict(ndc=True, near=0., far=1., N_samples=64, N_importance=64, perturb=1.0, raw_noise_std=0.)

    def step():
        rgb, rgb0, other, tens = model(400, 400, K, 1 << 22, rays=blur_rays, rays_info=None, force_naive=False, **kw)
        # the fused blur loss takes the per-sub-exposure colours and the weights; here the composed colours with unit weights
        ones = torch.ones((R, 1), device=dev)
        pb = blur_loss_partials_autograd(crf_rgb, rgb[:, None], ones, tgt, rgb0_p=rgb0[:, None], w2=ones, target_pts0=tgt0)
        if getattr(a, "merge_events", False):
            # the start and end rays of the event batch as ONE render call (a ray's colour does not depend on the other rays of its batch:
            # the same numbers as two calls up to the draw order of the stratification randoms); what a maintainer's integration can do
            # where run_nerf.py:534,547 calls nerf() twice -- fewer, larger launches
            s12, s120, _, _ = model(400, 400, K, 1 << 22, rays=torch.cat([ev_start, ev_end], 0), force_naive=True, tv=False, **kw)
            s1, s2, s10, s20 = s12[:E], s12[E:], s120[:E], s120[E:]
        else:
            s1, s10, _, _ = model(400, 400, K, 1 << 22, rays=ev_start, force_naive=True, tv=False, **kw)
            s2, s20, _, _ = model(400, 400, K, 1 << 22, rays=ev_end, force_naive=True, tv=False, **kw)
        pe = event_loss_partials_autograd(crf_ev, crf_flat, s1, s2, cum_neg, cum_pos, 0.2, 0.2, start0=s10, end0=s20, add_bii="pos-neg")
        if ddp:
            pb, pe = D.all_reduce_partials(pb, pe)
        loss, _ = blur_loss_from_partials(pb, fine_loss_weight=0.5, w_pts0=0.1)
        if "rgb_awp" in tens:               # the AWP composition's image term (run_nerf.py:470-480)
            loss = loss + ((tens["rgb_awp"] - tgt) ** 2).mean()
        loss = loss + 0.1 * event_loss_from_partials(pe) + 0.01 * other["TV"].sum()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        if ddp:
            ar_ev[0].record()
            red.start()
            red.wait()
            ar_ev[1].record()
        opt.step()
        crf_ev.load_params(crf_flat)
        if ddp:
            ar_ms.append(ar_ev)
        return loss

    for _ in range(3):
        l = step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    import contextlib
    # a.timed_ctx (tools/profile_train_kernels.py, tools/trace_h2d_profiler.py): a context manager entered around the TIMED iterations only --
    # the model construction above uploads 165 MB of grids in ~130 host-to-device copies, which a profile of the whole call would charge to
    # the iterations (round 3's "18 host-to-device copies per iteration" were exactly that)
    with (getattr(a, "timed_ctx", None) or contextlib.nullcontext()):
        e0.record()
        for _ in range(a.iters):
            l = step()
        e1.record()
        e1.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    if ddp:
        a.allreduce_ms = ar_ev[0].elapsed_time(ar_ev[1])            # the last iteration's gradient exchange AFTER loss.backward() returned: the exposed part
        a.allreduce_early_starts = red.early_starts
        a.grad_bytes = 4 * sum(p.numel() for p in list(model.parameters()) + [crf_flat])
    return ms, R * a.P + 2 * E, float(l.detach())


if __name__ == "__main__":
    main()
