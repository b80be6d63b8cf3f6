"""The float32-grade training mode (EVD_PREC_F16X3 in evd_nerf_mlp_train / evd_nerf_mlp_backward): every stored fragment a (hi, lo)
float16 pair, every product of the forward, dgrad and wgrad kernels the 3-MFMA split product.  The reference trains in float32
(run_nerf.py:593-601); in this mode the hand-written backward is held to float64 torch autograd with the TRUE ReLU pattern (no
"same pattern" allowance: a float32-grade forward does not flip units) and to the reference's own autograd goldens (G18) at 1e-3 of
the gradient norm -- the half-precision modes' bounds are 4e-3 (same pattern) / 15 % (golden)."""
import numpy as np
import pytest
import torch

from evdeblurnerf_amd import weights as W
from test_gpu_train import make_inputs, rel_l2
from torch_restatement import TorchNerf

pytestmark = pytest.mark.gpu


def test_f16x3_training_forward_is_the_inference_arithmetic():
    from evdeblurnerf_amd.nerf import NeRF
    sd = W.make_nerf_state_dict(21)
    for R, S in ((64, 64), (37, 9)):
        rb, z = make_inputs(R, S, 5)
        net = NeRF(sd, precision="f16x3")
        rbt, zt = torch.tensor(rb, device="cuda"), torch.tensor(z, device="cuda")
        raw, store = net.mlpforward_train(rbt, zt)
        assert torch.equal(raw, net.mlpforward(rbt, zt)[0])
        assert store.numel() == 2 * int(__import__("evdeblurnerf_amd")._lib.lib().evd_nerf_train_store_bytes(R * S))
        pts = torch.tensor(rb[:, None, 0:3] + rb[:, None, 3:6] * z[..., None], dtype=torch.float64).reshape(-1, 3)
        dirs = torch.tensor(np.repeat(rb[:, None, 8:11], S, 1), dtype=torch.float64).reshape(-1, 3)
        assert (raw.reshape(-1, 4).cpu().double() - TorchNerf(sd)(pts, dirs)).abs().max().item() < 2e-5


@pytest.mark.parametrize("R,S,gscale,tol", [(64, 64, 1e-4, 5e-6), (37, 9, 3.0, 5e-6), (300, 7, 1e-9, 5e-6), (512, 64, 1e-2, 5e-4)])
def test_f16x3_mlp_backward_matches_float64_autograd(R, S, gscale, tol):
    """All 24 parameter gradients vs float64 autograd of the restated network with ITS OWN ReLU pattern, d raw over 16 orders of
    magnitude across the cases (the loss scale).  Measured 4e-7 on the small cases; at 32 768 samples 1.2e-4 = one unit whose
    pre-activation is within float32 rounding of zero (float32 torch flips such units against float64 too)."""
    from evdeblurnerf_amd.nerf import NeRF
    sd = W.make_nerf_state_dict(22)
    rb, z = make_inputs(R, S, 6)
    rs = np.random.RandomState(9)
    d_raw = (rs.normal(size=(R, S, 4)) * gscale * np.exp(rs.uniform(-4, 0, (R, S, 1)))).astype(np.float32)
    net = NeRF(sd, precision="f16x3")
    raw, store = net.mlpforward_train(torch.tensor(rb, device="cuda"), torch.tensor(z, device="cuda"))
    grads = net.mlp_backward(torch.tensor(d_raw, device="cuda"), store)
    pts = torch.tensor(rb[:, None, 0:3] + rb[:, None, 3:6] * z[..., None], dtype=torch.float64).reshape(-1, 3)
    dirs = torch.tensor(np.repeat(rb[:, None, 8:11], S, 1), dtype=torch.float64).reshape(-1, 3)
    ref = TorchNerf(sd)
    (ref(pts, dirs) * torch.tensor(d_raw, dtype=torch.float64).reshape(-1, 4)).sum().backward()
    errs = {key: rel_l2(g.cpu().double(), ref.p[key.replace(".", "_")].grad) for key, g in grads.items()}
    print(f"[f16x3 R={R} S={S} g~{gscale:g}] worst relative L2 error of a parameter gradient vs float64 autograd: {max(errs.values()):.2e}")
    assert max(errs.values()) < tol, {k: f"{v:.1e}" for k, v in errs.items()}


def test_f16x3_gradients_reach_the_rays():
    from evdeblurnerf_amd.nerf import NeRF
    sd = W.make_nerf_state_dict(23)
    R, S = 48, 40
    rb_np, z_np = make_inputs(R, S, 12)
    wgt = (np.random.RandomState(13).normal(size=(R, S, 4)) * 1e-2).astype(np.float32)
    net = NeRF(sd, precision="f16x3")
    flat = net.flat_params(sd)
    rb = torch.tensor(rb_np, device="cuda", requires_grad=True)
    raw = net.mlp_train(flat, rb, torch.tensor(z_np, device="cuda"))
    (raw * torch.tensor(wgt, device="cuda")).sum().backward()
    rb64 = torch.tensor(rb_np, dtype=torch.float64, requires_grad=True)
    z64 = torch.tensor(z_np, dtype=torch.float64)
    # the sample positions as the reference computes them: pts = rays_o + rays_d * z in FLOAT32 (renderer.py:180) -- the encodings
    # multiply a position by up to 2^9, so float64 positions would move layer-0 pre-activations by ~1e-4 and flip units that the
    # reference's own arithmetic does not flip; the derivative still flows to the ray through the float64 expression
    pts32 = torch.tensor(rb_np[:, None, 0:3] + rb_np[:, None, 3:6] * z_np[..., None], dtype=torch.float64)
    expr = rb64[:, None, 0:3] + rb64[:, None, 3:6] * z64[..., None]
    pts = (pts32 + (expr - expr.detach())).reshape(-1, 3)
    dirs = rb64[:, None, 8:11].expand(-1, S, -1).reshape(-1, 3)
    ref = TorchNerf(sd)
    (ref(pts, dirs) * torch.tensor(wgt, dtype=torch.float64).reshape(-1, 4)).sum().backward()
    got = rb.grad.cpu().double()
    errs = {"rays_o": rel_l2(got[:, 0:3], rb64.grad[:, 0:3]), "rays_d": rel_l2(got[:, 3:6], rb64.grad[:, 3:6]),
            "viewdirs": rel_l2(got[:, 8:11], rb64.grad[:, 8:11]),
            "parameters": rel_l2(flat.grad.cpu().double(), torch.cat([ref.p[k.replace(".", "_")].grad.reshape(-1) for k, _, _ in net.param_blocks()]))}
    print("[f16x3] ray / parameter gradient relative L2 errors vs float64 autograd (true ReLU):", {k: f"{v:.1e}" for k, v in errs.items()})
    assert max(errs.values()) < 2e-4, errs


def test_f16x3_nerf_training_gradients_against_the_reference_golden():
    """G18 (w256 part): torch.autograd ON THE REFERENCE NeRF (8 x 256) + raw2outputs vs evd_nerf_mlp_train / _backward in the
    float32-grade mode + the compositing scan's backward kernel: every gradient's norm and seeded projection within 1e-3 of its
    norm (the half-precision test of the same golden accepts 15 %)."""
    from conftest import load_golden
    from evdeblurnerf_amd.nerf import NeRF
    from torch_restatement import grad_summary
    g = load_golden("G18_nerf_grads")
    sd = W.make_nerf_state_dict(19, D=8, W=256, rgb_add_bias=True)
    net = NeRF(sd, precision="f16x3").train()
    flat = net.flat_params(sd)
    o = torch.tensor(g["o"], device="cuda", requires_grad=True)
    d = torch.tensor(g["d"], device="cuda", requires_grad=True)
    z = torch.tensor(g["z"], device="cuda")
    R = o.shape[0]
    vd = d / d.norm(dim=-1, keepdim=True)
    rb = torch.cat([o, d, torch.zeros((R, 1), device="cuda"), torch.ones((R, 1), device="cuda"), vd], -1)
    raw = net.mlp_train(flat, rb, z)
    rgb_map = net.raw2outputs(raw, z, d)[0]
    assert np.abs(rgb_map.detach().cpu().numpy() - g["rgb_map_w256"]).max() < 2e-5
    (rgb_map * torch.tensor(g["w_rgb"], device="cuda")).sum().backward()
    got = dict(net.unflatten(flat.grad))
    got["rays_o"], got["rays_d"] = o.grad, d.grad
    keys = [k[5:-8] for k in g if k.startswith("w256.") and k.endswith(".summary")]
    assert set(keys) == set(got)
    worst = {}
    for idx, key in enumerate(keys):
        sm, _ = grad_summary(got[key].detach().cpu().numpy(), 7000 + idx)
        ref = g[f"w256.{key}.summary"]
        worst[key] = max(abs(sm[0] - ref[0]), abs(sm[1] - ref[1])) / float(ref[0])
    print("G18 (w256) vs the float32-grade kernels, worst (norm / projection error) / norm:",
          {k: f"{v:.1e}" for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:5]})
    assert max(worst.values()) < 1e-3, worst
