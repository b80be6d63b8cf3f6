"""GPU tests of the training engine's plumbing (VERDICT r2 item 4 / ADVICE r2): parameters as views of one flat storage per level,
the opt-in in-place gradient accumulation, the differentiable ray packing / sample positions as one launch each way."""
import numpy as np
import pytest
import torch

from evdeblurnerf_amd import weights as W
from test_gpu_train import _c2f_model

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rays(R, seed, requires_grad=True):
    return torch.tensor(W.synthetic_rays(seed, R), device=DEV, requires_grad=requires_grad)


def test_ray_batch_backward_matches_torch_autograd():
    """evd_ray_batch_bwd (viewdirs normalisation + NDC warp) vs torch.autograd of the same formula in float64 (renderer.py:423-446,
    utils/rays.py:104-145); forward vs evd_ray_batch's packing; also without the NDC warp."""
    from evdeblurnerf_amd.renderer import NeRFAll
    K = W.synthetic_camera()
    for ndc in (True, False):
        rays = _rays(777, 3)
        rb = NeRFAll.ray_batch_train(400, 400, K, rays, ndc=ndc)
        w = torch.randn((777, 11), device=DEV)
        (rb * w).sum().backward()
        r64 = rays.detach().double().cpu().requires_grad_(True)
        ref = NeRFAll.ray_batch_train(400, 400, K, r64, ndc=ndc)               # the torch arithmetic (CPU path of the same method)
        (ref * w.double().cpu()).sum().backward()
        assert (rb.detach().cpu().double() - ref.detach()).abs().max() < 2e-5
        err = (rays.grad.cpu().double() - r64.grad).abs().max() / r64.grad.abs().max()
        assert err < 2e-6, (ndc, float(err))


def test_points_backward_matches_torch_autograd():
    from evdeblurnerf_amd.renderer import points
    R, S = 333, 77
    rb = torch.randn((R, 11), device=DEV, requires_grad=True)
    z = torch.rand((R, S), device=DEV).sort(-1).values
    w = torch.randn((R, S, 3), device=DEV)
    pts = points(rb, z)
    (pts * w).sum().backward()
    rb2 = rb.detach().double().requires_grad_(True)
    ref = rb2[:, None, 0:3] + rb2[:, None, 3:6] * z.double()[..., None]
    (ref * w.double()).sum().backward()
    assert (pts.detach().double() - ref.detach()).abs().max() < 1e-6
    assert (rb.grad.double() - rb2.grad).abs().max() / rb2.grad.abs().max() < 1e-6
    assert points(rb.detach(), z).requires_grad is False


def _iteration(model, rays_list, K):
    kw = dict(ndc=True, near=0., far=1., use_viewdirs=True, N_samples=24, N_importance=16, raw_noise_std=0., perturb=0.)
    loss = 0.
    for rays in rays_list:                                          # three renders per iteration, like run_nerf.py:438,534,547
        rgb, rgb0, other, _ = model(400, 400, K, 1 << 22, rays=rays, **kw)
        loss = loss + (rgb ** 2).mean() + (rgb0 ** 2).mean() + 1e-3 * other["TV"].sum()
    return loss


def test_in_place_gradient_accumulation_equals_plain_autograd():
    """enable_training(grads_in_place=True) -- the backward kernels add into persistent buffers whose slices are the leaves' .grad --
    gives the gradients of the default mode (plain autograd returns) over an iteration of three renders + TV, also on the second
    iteration after optimizer.zero_grad() dropped them, and the optimizer's in-place update reaches the library (the next forward
    differs).  Default mode: torch.autograd.grad on net and grid leaves works."""
    K = W.synthetic_camera()
    rays_list = [_rays(256, s, requires_grad=False) for s in (1, 2, 3)]
    grads = {}
    for in_place in (False, True):
        model, sd = _c2f_model("f16", 16)
        model.enable_training(sd, grads_in_place=in_place).train()
        names = [n for n, _ in model.named_parameters()]
        opt = torch.optim.SGD(model.parameters(), lr=1e-2)
        per_iter = []
        for it in range(2):
            opt.zero_grad(set_to_none=True)
            loss = _iteration(model, rays_list, K)
            loss.backward()
            per_iter.append(([p.grad.detach().clone() for p in model.parameters()], float(loss)))
            opt.step()
        grads[in_place] = per_iter
        if not in_place:
            loss = _iteration(model, rays_list, K)
            g = torch.autograd.grad(loss, model.parameters(), allow_unused=True)
            assert all(x is not None and torch.isfinite(x).all() for x in g)
            assert sum(float(x.abs().sum()) for x in g) > 0
    for it in range(2):
        (ga, la), (gb, lb) = grads[False][it], grads[True][it]
        assert abs(la - lb) <= 2e-5 * abs(la), (it, la, lb)
        for n, a, b in zip(names, ga, gb):
            scale = float(a.abs().max()) + 1e-12
            # (the grid gradients are sums of float atomics: order-dependent rounding)
            assert float((a - b).abs().max()) <= 2e-3 * scale, (it, n, float((a - b).abs().max()), scale)
    assert grads[False][0][1] != grads[False][1][1]                 # the optimizer step changed what the library renders


def test_leaves_are_views_of_the_flat_storage_and_no_reupload_between_forwards():
    """The per-parameter leaves alias one flat tensor per level; a second forward without an optimizer step re-packs nothing
    (the level's sync keys -- address + shared version counter -- are unchanged), a step changes them."""
    model, sd = _c2f_model("f16", 16)
    model.enable_training(sd).train()
    lv = model._levels[1]
    base = lv.flat.data_ptr()
    offs = sorted(t.data_ptr() - base for t in lv.leaves.values())
    assert offs[0] == 0 and all(0 <= o < lv.flat.numel() * 4 for o in offs)
    K = W.synthetic_camera()
    rays = _rays(64, 5, requires_grad=False)
    kw = dict(ndc=True, near=0., far=1., use_viewdirs=True, N_samples=24, N_importance=16, raw_noise_std=0., perturb=0.)
    model(400, 400, K, 1 << 22, rays=rays, **kw)
    key_net, key_grid = model.mlp_fine._synced_net, model.mlp_fine._synced
    out = model(400, 400, K, 1 << 22, rays=rays, **kw)
    assert model.mlp_fine._synced_net == key_net and model.mlp_fine._synced == key_grid
    (out[0].sum() + out[1].sum()).backward()
    torch.optim.SGD(model.parameters(), lr=1e-2).step()
    model(400, 400, K, 1 << 22, rays=rays, **kw)
    assert model.mlp_fine._synced_net != key_net and model.mlp_fine._synced != key_grid
    sd2 = model.state_dict()
    assert set(sd2) == set(sd)                                      # the reference's keys, reference layouts
    for k in sd:
        assert tuple(sd2[k].shape) == tuple(np.asarray(sd[k]).shape), k


@pytest.mark.parametrize("in_place", [True, False])
def test_fused_adam_reaches_the_library(in_place):
    """torch.optim.Adam(fused=True) changes the parameters without bumping their version counters: the re-pack check must not depend on
    the counters alone (it also keys on "a backward ran since the last re-pack").  Two optimizers, same start, same data: the rendered
    colours after three steps must agree -- with stale packed copies the fused run would still render the initial network."""
    K = W.synthetic_camera()
    rays = _rays(256, 9, requires_grad=False)
    kw = dict(ndc=True, near=0., far=1., use_viewdirs=True, N_samples=24, N_importance=16, raw_noise_std=0., perturb=0.)
    tgt = torch.rand((256, 3), device=DEV, generator=torch.Generator(device=DEV).manual_seed(4))
    outs = []
    for fused in (False, True):
        model, sd = _c2f_model("f16x3", 16)
        model.enable_training(sd, grads_in_place=in_place).train()
        opt = torch.optim.Adam(model.parameters(), lr=2e-3, fused=fused)
        first = None
        for _ in range(3):
            rgb, rgb0, _, _ = model(400, 400, K, 1 << 22, rays=rays, tv=False, **kw)
            first = rgb.detach().clone() if first is None else first
            loss = ((rgb - tgt) ** 2).mean() + ((rgb0 - tgt) ** 2).mean()
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
        rgb = model(400, 400, K, 1 << 22, rays=rays, tv=False, **kw)[0].detach()
        assert (rgb - first).abs().max() > 1e-3                     # three steps moved the render
        outs.append(rgb)
    assert (outs[0] - outs[1]).abs().max() < 2e-4, float((outs[0] - outs[1]).abs().max())
