"""The WHOLE training call against the reference: golden G32 (NeRFAll.forward in training mode, networks/renderer.py:277-392, run by
tools/gen_golden.py with the reference's real RigidBlurringModel + AdaptiveWeightProposal, mode='c2f') and golden G33 (five iterations of
the optimisation loop of run_nerf.py:423-613 on that model).  The blur kernel is a PyTorch caller of the path and cannot travel to the GPU
box: a stub `kernelsnet` replays the outputs the real one produced (new_rays, weight, img_embed); everything behind it -- ray packing,
both PDRF levels, resampling, the adaptive weight proposal, the compositions, TV, the loss block, backward, Adam, re-pack -- is this
repository's path, called through `model(...)` with the reference call site's keyword set.

Tolerances are per mode and written next to the cases (G32_CASES, G33_CASES) with the values measured on MI355X.  In the float32-grade
mode f16x3: outputs 1e-5 on pixels whose P rays carry the golden's importance-sample positions (3e-5 on the others: sample_pdf's
conditioning, see tests/test_oracle_golden.py::test_G32_train_forward); (norm, seeded projection) of the level parameters' gradients
2.5e-4 of the norm in the median, 3e-3 on the worst tensor (grid gradients next to a moved sample); the loss of each of G33's five
iterations within 5e-6 of the reference's, the parameters after the fifth within 1e-3 of their norm."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import load_golden, maxabs, train_call_errors
from evdeblurnerf_amd import weights as W

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

AABB = ([-1.5, -1.5, -1.0], [1.5, 1.5, 1.0])
# render_kwargs_train of run_nerf.py:306-314 + the call site's own keys (:438-442)
CALL_KW = dict(retraw=True, perturb=0., N_importance=16, N_samples=16, use_viewdirs=True, white_bkgd=False, raw_noise_std=0., inference=False)


class ReplayKernel(torch.nn.Module):
    """stands where RigidBlurringModel stands (dpnerf/blurmodel.py:129-173): returns the recorded (new_rays, weight, align, extras) as
    autograd leaves, so the gradients the real kernel would receive can be read off them"""

    def __init__(self, g, prefix=""):
        super().__init__()
        self.g, self.prefix, self.step = g, prefix, 0
        self.last = None

    def forward(self, H, W_, K, rays, rays_info, feats=None, return_img_embed=False):
        p = self.prefix.format(self.step)
        leaf = lambda k: torch.tensor(self.g[p + k], device="cuda", requires_grad=True)
        self.last = dict(new_rays=leaf("new_rays"), weight=leaf("weight"), img_embed=leaf("img_embed"))
        assert rays.shape[0] == self.last["new_rays"].shape[0] and rays_info["images_idx"].shape[0] == rays.shape[0]
        return self.last["new_rays"], self.last["weight"], None, ({"img_embed": self.last["img_embed"]} if return_img_embed else {})


def _awp_module(seed, g, P):
    from awp_standin import RefLikeAWP
    awp = RefLikeAWP(P=P, view_ch=g["img_embed" if "img_embed" in g else "s0.img_embed"].shape[1], mam="corr")
    sd = {k[len("awp.sd."):]: torch.tensor(g[k]) for k in g if k.startswith("awp.sd.")}
    sd.update({k: torch.tensor(v) for k, v in W.make_awp_embed_state_dict(seed * 10 + 1).items()})
    missing, unexpected = awp.load_state_dict(sd, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    return awp.cuda().train()


def _model(seed, g, prec, awp_kind, kernel, grads_in_place=False):
    from evdeblurnerf_amd.awp import FusedAWP
    from evdeblurnerf_amd.renderer import NeRFAll
    gc, gf = [int(v) for v in g["grid_coarse"]], [int(v) for v in g["grid_fine"]]
    sd = W.make_train_call_state_dict(seed, gc, gf)
    P = g["weight" if "weight" in g else "s0.weight"].shape[1]
    args = SimpleNamespace(mode="c2f", multires=10, multires_views=4, use_viewdirs=True, N_importance=16, kernel_type="RBK", kernel_use_awp=True,
                           rgb_activate="sigmoid", sigma_activate="relu", bounding_box=AABB, coarse_num_layers=2, coarse_num_layers_color=3,
                           coarse_hidden_dim=64, coarse_hidden_dim_color=64, coarse_app_dim=32, coarse_app_n_comp=[64, 16, 16], coarse_n_voxels=24 ** 3,
                           kernel_feat_cnl=15, fine_num_layers=2, fine_num_layers_color=3, fine_hidden_dim=256, fine_hidden_dim_color=256,
                           fine_geo_feat_dim=128, fine_app_dim=32, fine_app_n_comp=[64, 16, 16], fine_n_voxels=48 ** 3)
    awp = _awp_module(seed, g, P)
    awpnet = FusedAWP(awp, precision="bf16" if prec == "bf16" else "f16") if awp_kind == "fused" else awp
    model = NeRFAll(args, sd, kernelsnet=kernel, awpnet=awpnet, precision=prec).enable_training(sd, grads_in_place=grads_in_place).train()
    assert model.use_awp and model.mlp_coarse.gridSize == gc and model.mlp_fine.gridSize == gf
    return model, awp, sd


def _ref_layout(k, grad):
    if ".app_plane." in k:
        return grad.permute(2, 0, 1).unsqueeze(0)                       # channel-last [H,W,C] -> the reference's [1,C,H,W]
    if ".app_line." in k:
        return grad.t().unsqueeze(0).unsqueeze(-1)
    return grad


def rel(a, b):
    a, b = np.asarray(a, np.float64).reshape(-1), np.asarray(b, np.float64).reshape(-1)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


# tolerances per (precision, awp kind): outputs on the pixels that carry the golden's sample positions / on all pixels, rgb_awp, the level
# gradients' (norm, projection) error of the norm -- median over the 24 tensors and worst tensor --, AWP-side / kernel-side gradients.
# Measured on MI355X (profiles/r06_train_call_parity.log): f16x3 + torch AWP 7e-7 / 1.0e-5 / 1.5e-4 / 1.4e-3 / 7e-3; f16m 3.6e-4 / 1.5e-3;
# f16c 4.5e-4 / 1.4e-3; f16 8.7e-3 / 5.1e-2.  The worst level tensors are grid gradients (fine x-y plane, coarse x line): a third of the
# rays carry an importance sample that sits 1e-4 away from the reference's (sample_pdf's conditioning), which moves a tri-plane tap's weights.
G32_CASES = {
    ("f16x3", "torch"): dict(out=1e-5, out_all=3e-5, awp_out=1e-5, level_med=2.5e-4, level=3e-3, side=2e-2),
    ("f16x3", "fused"): dict(out=1e-5, out_all=3e-5, awp_out=1e-5, level_med=2.5e-4, level=3e-3, side=4e-2),
    ("f16m", "fused"): dict(out=1e-5, out_all=3e-5, awp_out=1e-5, level_med=8e-4, level=4e-3, side=4e-2),
    ("f16c", "fused"): dict(out=5e-5, out_all=5e-5, awp_out=5e-5, level_med=1e-3, level=4e-3, side=4e-2),
    ("f16", "fused"): dict(out=5e-4, out_all=5e-4, awp_out=5e-4, level_med=2.5e-2, level=0.15, side=0.1),
}


@pytest.mark.parametrize("prec,awp_kind", list(G32_CASES))
def test_G32_training_call_matches_the_reference(prec, awp_kind):
    from torch_restatement import check_grad_elements, grad_summary
    tol = G32_CASES[(prec, awp_kind)]
    g = load_golden("G32_train_forward")
    kern = ReplayKernel(g)
    model, awp, sd = _model(32, g, prec, awp_kind, kern)
    R, P = g["weight"].shape
    seen = {}
    inner = model.awpnet
    orig = inner.forward
    inner.forward = lambda df, z, rd, vf: (seen.update(z_vals=z.detach().reshape(R * P, -1).cpu().numpy(), rays_d=rd.detach().reshape(R * P, 3).cpu().numpy()), orig(df, z, rd, vf))[1]
    K = W.synthetic_camera()
    rays = torch.tensor(g["rays"], device="cuda")
    info = {"images_idx": torch.tensor(g["images_idx"], device="cuda")}
    rgb, rgb1, other_loss, other_tensors = model(400, 400, K, 1 << 20, rays=rays, rays_info=info, force_naive=False, return_pts0_rgb=True, **CALL_KW)
    # the reference's key sets (renderer.py:347,361-376)
    assert set(other_loss) == {"TV"} and set(other_tensors) == {"rgb_awp", "stage1_img_embed", "stage1_rgb_pts0", "stage1_rgb1_pts0"}
    assert torch.equal(other_tensors["stage1_img_embed"], kern.last["img_embed"])
    out = dict(rgb=rgb, rgb1=rgb1, rgb_awp=other_tensors["rgb_awp"], stage1_rgb_pts0=other_tensors["stage1_rgb_pts0"],
               stage1_rgb1_pts0=other_tensors["stage1_rgb1_pts0"])
    assert maxabs(seen["rays_d"], g["awp_in_rays_d"]) < 2e-6                      # the AWP receives the NDC directions (renderer.py:464-465)
    tight, e_t, e_a = train_call_errors({k: v.detach().cpu().numpy() for k, v in out.items()}, seen["z_vals"], g, z_tol=5e-6 if prec in ("f16x3", "f16m") else 2e-4)
    print(f"G32 [{prec}, {awp_kind} AWP] outputs: {int(tight.sum())} of {R} pixels on the golden's sample positions;", {k: f"{v:.1e}" for k, v in e_t.items()},
          "all pixels:", {k: f"{v:.1e}" for k, v in e_a.items()})
    loss = sum((out[k] * torch.tensor(g["proj." + k], device="cuda")).sum() for k in out) + 0.1 * other_loss["TV"].sum()
    loss.backward()
    # ---- gradients of the level parameters: (norm, projection) + element pins
    got = {k: _ref_layout(k, v.grad) for k, v in model.named_parameters() if k.startswith(("mlp_coarse.", "mlp_fine."))}
    keys = [k[2:-8] for k in g if k.startswith("g.mlp_") and k.endswith(".summary")]
    assert set(keys) == set(got), set(keys) ^ set(got)
    worst, elem = {}, {}
    for idx, key in enumerate(keys):
        a = got[key].detach().cpu().numpy()
        sm, _ = grad_summary(a, 7000 + idx)
        ref = g[f"g.{key}.summary"]
        worst[key] = max(abs(sm[0] - ref[0]), abs(sm[1] - ref[1])) / float(ref[0])
        elem[key], _ = check_grad_elements(a, g[f"g.{key}.elem_idx"], g[f"g.{key}.elem_val"], 1.0)
    top = lambda d: {k: f"{v:.1e}" for k, v in sorted(d.items(), key=lambda kv: -kv[1])[:5]}
    print(f"G32 [{prec}, {awp_kind} AWP] level gradients, (norm / projection error) / norm: median {np.median(list(worst.values())):.1e}, worst", top(worst))
    print(f"G32 [{prec}, {awp_kind} AWP] level gradients, worst pinned element / largest:", top(elem))
    # ---- the AWP's parameters and what the blur kernel receives
    side = {}
    for k, p in awp.named_parameters():
        if "g.awp." + k not in g:
            continue
        ref = g["g.awp." + k]
        if k == "MAM.linear.bias":                   # analytically zero (the training-mode BatchNorm removes a constant added to every curve)
            assert p.grad is None or p.grad.abs().max().item() < 1e-4
            continue
        assert p.grad is not None, k
        side["awp." + k] = rel(p.grad.cpu().numpy().reshape(ref.shape), ref)
    for k in ("new_rays", "weight", "img_embed"):
        side[k] = rel(kern.last[k].grad.cpu().numpy(), g["g." + k])
    print(f"G32 [{prec}, {awp_kind} AWP] AWP / kernel-side gradients, error / norm:", top(side))
    assert tight.sum() >= 6
    assert max(v for k, v in e_t.items() if k != "rgb_awp") < tol["out"] and e_t["rgb_awp"] < tol["awp_out"], e_t
    assert max(e_a.values()) < tol["out_all"], e_a
    assert abs(other_loss["TV"].item() - float(g["tv"])) < 1e-4 * float(g["tv"])
    assert abs(loss.item() - float(g["loss"])) < 200 * tol["out_all"]
    assert np.median(list(worst.values())) < tol["level_med"] and max(worst.values()) < tol["level"], top(worst)
    assert max(side.values()) < tol["side"], top(side)
    if awp_kind == "fused":                            # BatchNorm running estimates after the step (the fused tail updates the wrapped module's buffers)
        bn = awp.MAM.Corr.convd[1]
        assert maxabs(bn.running_mean.cpu().numpy(), g["awp.after.running_mean"]) < 2e-3 and int(bn.num_batches_tracked) == 1


# ------------------------------------------------------------------------------------------------------------ G33: five iterations
# per (precision, awp kind, grads_in_place): bound on |loss - reference loss| per step, on the worst (norm / projection) error of a
# level tensor's CHANGE after the last step relative to the change's norm, the same for the AWP / CRF tensors, and on the parameters
# themselves (error of the value / norm of the value)
G33_CASES = {       # measured (profiles/r06_train_call_parity.log): loss 9e-7 ... 7e-6; level 1.1e-2 (f16x3, f16m, f16c: 9e-3) / 5.4e-2 (f16); value 4e-4 ... 2e-3
    ("f16x3", "torch", False): dict(loss=5e-6, level=3e-2, side=3e-2, value=1e-3),
    ("f16x3", "fused", True): dict(loss=5e-6, level=3e-2, side=0.15, value=5e-3),
    ("f16m", "fused", True): dict(loss=1e-5, level=3e-2, side=0.15, value=5e-3),
    ("f16c", "fused", True): dict(loss=1e-5, level=3e-2, side=0.15, value=5e-3),
    ("f16", "fused", True): dict(loss=5e-5, level=0.15, side=0.2, value=6e-3),
}


@pytest.mark.parametrize("prec,awp_kind,in_place", list(G33_CASES))
def test_G33_five_iterations_follow_the_reference_trajectory(prec, awp_kind, in_place):
    """run_nerf.py:423-613 on the G32 model: blur batch through the (replayed) kernel + AWP, event batch's start / end rays, the loss
    block on the fused reductions, backward, Adam over the reference's groups {grad_vars, grad_vars_vol, crf}, the decay of :603-613,
    re-pack.  Compared after every step: the loss and the CHANGE of every level / AWP / event-CRF parameter (Adam's first steps have
    the size of the learning rate whatever the gradient's size, so an element whose gradient is rounding noise around zero moves by a full
    +- lr on either side: the bounds are on norms and projections of whole tensors, not on single elements)."""
    from torch_restatement import check_grad_elements, grad_summary
    from evdeblurnerf_amd.losses import (blur_loss_partials_autograd, crf_param_grads, event_loss_from_partials, event_loss_partials_autograd)
    from evdeblurnerf_amd.tonemapping import CRF
    tol = G33_CASES[(prec, awp_kind, in_place)]
    g = load_golden("G33_train_trajectory")
    lrate, lrate_decay, flw, w_pts0, w_egm, w_tv, thr = (float(v) for v in g["scalars"])
    STEPS = len(g["losses"])
    kern = ReplayKernel(g, prefix="s{}.")
    model, awp, sd = _model(33, g, prec, awp_kind, kern, grads_in_place=in_place)
    csd = W.make_crf_state_dict(331, extra_features=2)
    csd = {k: (v * np.float32(3.0) if np.asarray(v).ndim == 2 else v) for k, v in csd.items()}
    crf_rgb, crf_ev = CRF("gamma"), CRF("learn", state_dict=csd, extra_features=2)
    crf_flat = crf_ev.flat_params("cuda")
    groups = [{"params": model.grad_vars, "lr": lrate}, {"params": model.grad_vars_vol, "lr": lrate}, {"params": [crf_flat], "lr": lrate}]
    for gr in groups:
        gr.setdefault("initial_lr", gr["lr"])
    assert {id(p) for p in model.grad_vars + model.grad_vars_vol} == {id(p) for p in model.parameters()}
    opt = torch.optim.Adam(params=groups, lr=lrate, betas=(0.9, 0.999))
    dev = "cuda"
    K = W.synthetic_camera()
    T = lambda k: torch.tensor(g[k], device=dev)
    rays, ev_start, ev_end, target, target_pts0, cn, cp = (T(k) for k in ("rays", "ev_start", "ev_end", "target", "target_pts0", "cn", "cp"))
    info = {"images_idx": T("images_idx")}
    R = rays.shape[0]
    ones = torch.ones((R, 1), device=dev)

    def tracked():
        out = {k: _ref_layout(k, v.detach()) for k, v in model.named_parameters() if k.startswith(("mlp_coarse.", "mlp_fine."))}
        out.update({"awp." + k: v.detach() for k, v in awp.named_parameters() if not k.startswith("MAM.conv.")})
        out.update({"crf." + k: v for k, v in crf_param_grads(crf_flat.detach(), 2).items()})
        return {k: v.cpu().numpy().astype(np.float64) for k, v in out.items()}

    p0 = tracked()
    keys = [k[len("s0.d."):-8] for k in g if k.startswith("s0.d.") and k.endswith(".summary")]
    assert set(keys) == set(p0), set(keys) ^ set(p0)
    global_step, losses, report = 0, [], {}
    for i in range(STEPS):
        kern.step = i
        rgb, rgb0, other, tens = model(400, 400, K, 1 << 20, rays=rays, rays_info=info, force_naive=False, return_pts0_rgb=True, **CALL_KW)
        pa = blur_loss_partials_autograd(crf_rgb, rgb[:, None], ones, target, rgb0_p=rgb0[:, None])
        pb = blur_loss_partials_autograd(crf_rgb, tens["rgb_awp"][:, None], ones, target)
        pc = blur_loss_partials_autograd(crf_rgb, tens["stage1_rgb_pts0"][:, None], ones, target_pts0, rgb0_p=tens["stage1_rgb1_pts0"][:, None])
        n = pa.detach()[5]
        loss = (pa[0] + pa[1]) / n * (1 - flw) + pb[0] / n * flw + (pc[0] + pc[1]) / n * w_pts0
        loss = loss + other["TV"].mean() * w_tv
        s, s0, _, _ = model(400, 400, K, 1 << 20, rays=ev_start, rays_info=None, force_naive=True, **CALL_KW)
        e, e0, _, _ = model(400, 400, K, 1 << 20, rays=ev_end, rays_info=None, force_naive=True, **CALL_KW)
        pe = event_loss_partials_autograd(crf_ev, crf_flat, s, e, cn, cp, thr, thr, start0=s0, end0=e0, add_bii="pos-neg")
        loss = loss + event_loss_from_partials(pe) * w_egm
        opt.zero_grad()
        loss.backward()
        opt.step()
        crf_ev.load_params(crf_flat)
        for gr in opt.param_groups:                     # run_nerf.py:603-613
            gr["lr"] = gr["initial_lr"] * (0.1 ** (global_step / (lrate_decay * 1000)))
        global_step += 1
        losses.append(float(loss.detach()))
        now = tracked()
        worst = {}
        for idx, key in enumerate(keys):
            if key == "awp.MAM.linear.bias":            # its gradient is analytically zero: both sides step along rounding noise
                continue
            sm, _ = grad_summary(now[key] - p0[key], 7000 + idx)
            ref = g[f"s{i}.d.{key}.summary"]
            worst[key] = max(abs(sm[0] - ref[0]), abs(sm[1] - ref[1])) / max(float(ref[0]), 1e-30)
        report[i] = worst
    top = lambda d: {k: f"{v:.1e}" for k, v in sorted(d.items(), key=lambda kv: -kv[1])[:4]}
    lerr = np.abs(np.array(losses) - g["losses"])
    print(f"G33 [{prec}, {awp_kind} AWP, in_place={in_place}] losses", [f"{v:.6f}" for v in losses], "reference", [f"{v:.6f}" for v in g["losses"]], f"max diff {lerr.max():.1e}")
    for i in range(STEPS):
        lv = {k: v for k, v in report[i].items() if k.startswith("mlp_")}
        sd_ = {k: v for k, v in report[i].items() if not k.startswith("mlp_")}
        print(f"  step {i}: change of the level tensors, (norm / projection error) / norm: median {np.median(list(lv.values())):.1e} worst {top(lv)}; AWP / CRF: median {np.median(list(sd_.values())):.1e} worst {top(sd_)}")
    last = tracked()
    value = {}
    for key in keys:
        if f"s{STEPS - 1}.p.{key}" in g and key != "awp.MAM.linear.bias":
            ref = g[f"s{STEPS - 1}.p.{key}"].astype(np.float64)
            value[key] = float(np.linalg.norm(last[key].reshape(ref.shape) - ref) / max(np.linalg.norm(ref), 1e-30))
    print(f"  parameter values after step {STEPS - 1} (small tensors in full), error / norm: worst {top(value)}")
    assert np.all(np.isfinite(losses)) and losses[-1] < losses[0]
    assert lerr.max() < tol["loss"], lerr
    lv = {k: v for k, v in report[STEPS - 1].items() if k.startswith("mlp_")}
    sd_ = {k: v for k, v in report[STEPS - 1].items() if not k.startswith("mlp_")}
    assert max(lv.values()) < tol["level"], top(lv)
    assert max(sd_.values()) < tol["side"], top(sd_)
    assert max(value.values()) < tol["value"], top(value)
    bn = awp.MAM.Corr.convd[1]
    assert int(bn.num_batches_tracked) == int(g["awp.after.num_batches_tracked"]) == STEPS
    assert maxabs(bn.running_mean.cpu().numpy(), g["awp.after.running_mean"]) < 5e-3


def test_parameters_are_the_reference_models_parameters():
    """rgb_add_bias off (every shipped config): the reference's colour networks have NO bias tensors (voxnerf.py:80 `bias=add_bias_color`),
    so none may appear in named_parameters() / the optimizer groups / state_dict() -- the library's flat layout always has the slots, and
    exposing them trained six extra vectors per model that the reference does not have (found by G32)."""
    g = load_golden("G32_train_forward")
    model, awp, sd = _model(32, g, "f16", "fused", ReplayKernel(g))
    names = [k for k, _ in model.named_parameters()]
    assert not any(k.endswith("color_net.0.bias") or k.endswith("color_net.1.bias") or k.endswith("color_net.2.bias") for k in names)
    assert {k for k in names if k.startswith("mlp_")} == set(sd)
    assert {k for k in model.state_dict() if k.startswith("mlp_")} == set(sd)
    assert {k for k in names if k.startswith("awpnet.")} == {"awpnet." + k for k, _ in awp.named_parameters()}
