"""The per-ray remainder of the adaptive weight proposal on the GPU (evd_awp_tail_forward / _backward; reference networks/dpnerf/awp.py:89-95,
104-117 and networks/dpnerf/mam.py:35-53) against golden G27 (the reference's AdaptiveWeightProposal with its real MotionAggregationModule,
training mode, torch.autograd gradients), against the C oracle at the shipped sizes, and against float64 autograd of tools/awp_standin.py
(pinned to G27 on the CPU side, tests/test_oracle_golden.py) at sizes where a workgroup walks several rays.

Tolerances: everything is float32 on both sides; outputs within 2e-6 (they are probabilities of order 0.1), gradients within 1e-4 of the
tensor's norm (the sums over R x S samples are accumulated in a different order than torch's)."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def rel(a, b):
    a, b = torch.as_tensor(a).double().reshape(-1).cpu(), torch.as_tensor(b).double().reshape(-1).cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _standin_from_golden(g):
    from awp_standin import RefLikeAWP
    P, VF = g["out"].shape[1], g["view_feature"].shape[1]
    awp = RefLikeAWP(P=P, view_ch=VF, mam="corr")
    missing, unexpected = awp.load_state_dict({k[3:]: torch.tensor(g[k]) for k in g if k.startswith("sd.")}, strict=False)
    assert not unexpected
    return awp.cuda()


def _kernel_chain(fused, h_local, z, rays_d, view_feature, R, P, S):
    """integration -> mam_local -> the tail kernels, as FusedAWP.forward chains them behind the embedding"""
    from evdeblurnerf_amd.awp import feature_integration, mam_local
    mam = fused.ref.MAM
    h = feature_integration(h_local.reshape(R, P, S, -1), z, rays_d)
    h_inter, h_intra = mam_local(h_local, mam.linear.weight, mam.Corr.line_conv_att.weight, R, P, S)
    out = fused._tail(h, view_feature, rays_d, h_inter, h_intra, R, P, S)
    assert out is not None, "the library refused a shape it is built for"
    return out


def test_tail_matches_golden_G27():
    from evdeblurnerf_amd.awp import FusedAWP
    g = load_golden("G27_awp_per_ray")
    R, P = g["out"].shape
    S = g["z"].shape[1]
    awp = _standin_from_golden(g).train()
    fused = FusedAWP(awp)
    assert fused.tail_kernels and fused._F == 2
    hl, rd, vf = (torch.tensor(g[k]).cuda().requires_grad_(True) for k in ("h_local", "rays_d", "view_feature"))
    out = _kernel_chain(fused, hl, torch.tensor(g["z"]).cuda(), rd, vf, R, P, S)
    assert (out.detach().cpu() - torch.tensor(g["out"])).abs().max().item() < 5e-6      # (float32 golden, BatchNorm over 28 rows: its own rounding is 2e-6)
    names = [k[2:] for k in g if k.startswith("g.") and k[2:] not in ("h_local", "rays_d", "view_feature")]
    pd = dict(awp.named_parameters())
    grads = torch.autograd.grad((out * torch.tensor(g["proj"]).cuda()).sum(), [hl, rd, vf] + [pd[k] for k in names])
    worst = {}
    for got, key in zip(grads, ["h_local", "rays_d", "view_feature"] + names):
        ref = g["g." + key]
        if key == "MAM.linear.bias":         # analytically zero (the training-mode BatchNorm removes a constant added to every curve)
            assert got.abs().max().item() < 1e-4
            continue
        worst[key] = rel(got.reshape(ref.shape), ref)
    print("G27 gradient errors (of the norm):", {k: f"{v:.1e}" for k, v in worst.items()})
    assert max(worst.values()) < 1e-4, worst
    bn = awp.MAM.Corr.convd[1]
    assert (bn.running_mean.cpu() - torch.tensor(g["after.running_mean"])).abs().max().item() < 1e-6
    assert (bn.running_var.cpu() - torch.tensor(g["after.running_var"])).abs().max().item() < 1e-6
    assert int(bn.num_batches_tracked) == int(g["after.num_batches_tracked"])
    awp.eval()
    with torch.no_grad():
        out_e = _kernel_chain(fused, hl.detach(), torch.tensor(g["z"]).cuda(), rd.detach(), vf.detach(), R, P, S)
    assert (out_e.cpu() - torch.tensor(g["out_eval"])).abs().max().item() < 2e-6
    assert int(bn.num_batches_tracked) == int(g["after.num_batches_tracked"])        # eval: the estimates are read, not updated


def _random_case(R, P, S, VF, seed, n_extra_layers=0):
    from awp_standin import RefLikeAWP
    torch.manual_seed(seed)
    awp = RefLikeAWP(P=P, view_ch=VF, mam="corr")
    for _ in range(n_extra_layers):
        awp.motion_feature_embed_layer.append(torch.nn.Linear(32, 32))
    corr = awp.MAM.Corr
    with torch.no_grad():                            # attention logits of order 1, a BatchNorm away from its initial values
        for conv, k in ((corr.conva, 6.0), (corr.convb, 6.0), (corr.convc, 6.0), (corr.line_conv_att, 20.0)):
            conv.weight.mul_(k)
        corr.convd[1].weight.uniform_(0.5, 1.5)
        corr.convd[1].bias.normal_(0, 0.2)
    rs = np.random.RandomState(seed)
    h_local = np.maximum(rs.standard_normal((R * P, S, 64)), 0).astype(np.float32) * 0.5
    z = np.sort(rs.uniform(0, 1, (R * P, S)).astype(np.float32), -1)
    rays_d = rs.standard_normal((R * P, 3)).astype(np.float32)
    vf = rs.standard_normal((R, VF)).astype(np.float32) if VF else None
    proj = rs.standard_normal((R, P)).astype(np.float32)
    return awp, h_local, z, rays_d, vf, proj


@pytest.mark.parametrize("R,P,S,VF,extra", [(300, 10, 128, 32, 0), (37, 5, 64, 4, 1), (9, 16, 40, 0, 0), (530, 10, 128, 32, 0)])
def test_tail_vs_float64_autograd(R, P, S, VF, extra):
    """the shipped shape (P 10, S 64 + 64, view embedding 32 + 15 direction columns) with more rays than workgroups (a workgroup walks 2-3
    rays; the finish kernels' last block is partly empty), a small odd one with a second hidden layer, P = 16 without view_feature"""
    import copy
    from evdeblurnerf_amd.awp import FusedAWP
    awp, h_local, z, rays_d, vf, proj = _random_case(R, P, S, VF, seed=R + P, n_extra_layers=extra)
    ref = copy.deepcopy(awp).double().train()
    awp = awp.cuda().train()
    fused = FusedAWP(awp)
    assert fused.tail_kernels
    dev = lambda a, dt=torch.float32: None if a is None else torch.tensor(a, dtype=dt).cuda().requires_grad_(True)
    hl, rd, v = dev(h_local), dev(rays_d), dev(vf)
    out = _kernel_chain(fused, hl, torch.tensor(z).cuda(), rd, v, R, P, S)
    ins = [hl, rd] + ([v] if v is not None else [])
    pk = [p_ for n_, p_ in awp.named_parameters() if not n_.startswith(("sample_feature_embed_layer", "MAM.conv."))]
    grads = torch.autograd.grad((out * torch.tensor(proj).cuda()).sum(), ins + pk)
    c64 = lambda a: None if a is None else torch.tensor(a, dtype=torch.float64, requires_grad=True)
    hl64, rd64, v64 = c64(h_local), c64(rays_d), c64(vf)
    out64 = ref.forward_from_local(hl64, torch.tensor(z, dtype=torch.float64), rd64, v64)
    ins64 = [hl64, rd64] + ([v64] if v64 is not None else [])
    names = [n_ for n_, _ in ref.named_parameters() if not n_.startswith(("sample_feature_embed_layer", "MAM.conv."))]
    pr = dict(ref.named_parameters())
    grads64 = torch.autograd.grad((out64 * torch.tensor(proj, dtype=torch.float64)).sum(), ins64 + [pr[n_] for n_ in names])
    err_out = (out.detach().cpu().double() - out64.detach()).abs().max().item()
    assert err_out < 2e-6, err_out
    keys = ["h_local", "rays_d"] + (["view_feature"] if v is not None else []) + names
    worst = {}
    for key, a, b in zip(keys, grads, grads64):
        if key == "MAM.linear.bias":
            assert a.abs().max().item() < 1e-4 * max(1.0, grads64[0].abs().max().item())
            continue
        worst[key] = rel(a, b)
    print(f"[R {R} P {P} S {S} VF {VF}] out {err_out:.1e}; gradients:", {k: f"{v_:.1e}" for k, v_ in worst.items()})
    assert max(worst.values()) < 1e-4, worst
    bn, bn64 = awp.MAM.Corr.convd[1], ref.MAM.Corr.convd[1]
    assert (bn.running_mean.cpu().double() - bn64.running_mean).abs().max().item() < 1e-6
    assert (bn.running_var.cpu().double() - bn64.running_var).abs().max().item() < 1e-6


def test_tail_forward_matches_oracle_at_the_shipped_shape(O):
    """the C oracle (evo_awp_feature_integration -> evo_mam_local -> evo_awp_per_ray, the reference as written) on 256 rays x 10 x 128"""
    from evdeblurnerf_amd.awp import FusedAWP
    R, P, S, VF = 256, 10, 128, 32
    awp, h_local, z, rays_d, vf, _ = _random_case(R, P, S, VF, seed=5)
    sd = {k: v.detach().numpy() for k, v in awp.state_dict().items()}
    h = O.awp_feature_integration(h_local, z, rays_d).reshape(R, P, -1)
    d0 = rays_d.reshape(R, P, 3)[:, 0]
    view = np.concatenate([vf, O.embed(d0 / np.linalg.norm(d0, axis=-1, keepdims=True), 2)], -1)
    inter, intra = O.mam_local(h_local, sd["MAM.linear.weight"], sd["MAM.linear.bias"], sd["MAM.Corr.line_conv_att.weight"], P)
    want, mean, var = O.awp_per_ray(h, view, inter, intra, sd, training=True)
    awp = awp.cuda().train()
    fused = FusedAWP(awp)
    with torch.no_grad():
        out = _kernel_chain(fused, torch.tensor(h_local).cuda(), torch.tensor(z).cuda(), torch.tensor(rays_d).cuda(), torch.tensor(vf).cuda(), R, P, S)
    assert np.abs(out.cpu().numpy() - want).max() < 2e-6
    bn = awp.MAM.Corr.convd[1]
    assert np.abs(bn.running_mean.cpu().numpy() - 0.1 * mean).max() < 1e-6             # from 0 / 1 with momentum 0.1
    assert np.abs(bn.running_var.cpu().numpy() - (0.9 + 0.1 * var)).max() < 1e-6


def test_fused_awp_whole_forward_kernels_vs_torch_remainder():
    """FusedAWP end to end (embedding on float32 rows) with the remainder on the kernels and on the channel-last torch path: same weights
    and gradients; a sample count the LDS cannot hold falls back without an error"""
    import copy
    from awp_standin import RefLikeAWP
    from evdeblurnerf_amd.awp import FusedAWP
    torch.manual_seed(3)
    R, P, S = 96, 10, 128
    base = RefLikeAWP(P=P, view_ch=32, mam="corr").cuda().train()
    rs = np.random.RandomState(4)
    df = torch.tensor((rs.standard_normal((R * P, S, 128)) * 0.7).astype(np.float32)).cuda()
    z = torch.tensor(np.sort(rs.uniform(0, 1, (R * P, S)).astype(np.float32), -1)).cuda()
    rd = torch.tensor(rs.standard_normal((R * P, 3)).astype(np.float32)).cuda().requires_grad_(True)
    vf = torch.tensor(rs.standard_normal((R, 32)).astype(np.float32)).cuda().requires_grad_(True)
    proj = torch.tensor(rs.standard_normal((R, P)).astype(np.float32)).cuda()
    res = {}
    for tag, kern in (("kernels", True), ("torch", False)):
        m = copy.deepcopy(base)
        fused = FusedAWP(m, precision="f16", tail_kernels=kern)
        assert fused.tail_kernels == kern
        out = fused(df, z, rd, vf)
        ps = [p_ for n_, p_ in m.named_parameters() if not n_.startswith("MAM.conv.")]
        res[tag] = (out.detach(), torch.autograd.grad((out * proj).sum(), [rd, vf] + ps), m.MAM.Corr.convd[1].running_var.clone())
    assert (res["kernels"][0] - res["torch"][0]).abs().max().item() < 5e-6
    names = ["rays_d", "view_feature"] + [n_ for n_, _ in base.named_parameters() if not n_.startswith("MAM.conv.")]
    for key, a, b in zip(names, res["kernels"][1], res["torch"][1]):
        if key == "MAM.linear.bias":
            continue
        assert rel(a, b) < (3e-3 if key.startswith("sample_feature_embed_layer") else 2e-4), (key, rel(a, b))   # (the f16 embedding's own rounding)
    assert (res["kernels"][2] - res["torch"][2]).abs().max().item() < 1e-6
    big = FusedAWP(copy.deepcopy(base), precision="f16")
    S2 = 384
    df2 = torch.tensor((rs.standard_normal((8 * P, S2, 128)) * 0.7).astype(np.float32)).cuda()
    z2 = torch.tensor(np.sort(rs.uniform(0, 1, (8 * P, S2)).astype(np.float32), -1)).cuda()
    out2 = big(df2, z2, rd[:8 * P].detach(), vf[:8].detach())
    assert out2.shape == (8, P) and (P, S2, 32, 2, 2) in big._tail_refused
    assert (out2.sum(-1) - 1).abs().max().item() < 1e-5


@pytest.mark.parametrize("variant", ["eval_mode", "one_layer", "four_layers", "foreign_encoding"])
def test_tail_variants_vs_float64_autograd(variant):
    """eval mode (the BatchNorm's running estimates are constants of the backward), a motion embedding of one and of four layers
    (kernel_awp_mot_emb_depth 0 / 3), and a direction encoding the kernel does not know (its columns then arrive as view_feature)"""
    import copy
    from evdeblurnerf_amd.awp import FusedAWP
    R, P, S, VF = 70, 6, 48, 7
    awp, h_local, z, rays_d, vf, proj = _random_case(R, P, S, VF, seed=77, n_extra_layers=2 if variant == "four_layers" else 0)
    if variant == "one_layer":
        awp.motion_feature_embed_layer = torch.nn.ModuleList([awp.motion_feature_embed_layer[0]])
    if variant == "foreign_encoding":       # three frequencies, cosines first: not get_embedder's layout
        awp.ray_dirs_embed_fn = lambda x: torch.cat([x] + [f(x * 2.0 ** k) for k in range(2) for f in (torch.cos, torch.sin)], -1)
    with torch.no_grad():
        bn = awp.MAM.Corr.convd[1]
        bn.running_mean.normal_(0, 0.05)
        bn.running_var.uniform_(0.01, 0.05)
    ref = copy.deepcopy(awp).double()
    awp = awp.cuda()
    (awp.eval(), ref.eval()) if variant == "eval_mode" else (awp.train(), ref.train())
    fused = FusedAWP(awp)
    assert fused.tail_kernels and (fused._F is None) == (variant == "foreign_encoding")
    dev = lambda a: torch.tensor(a).cuda().requires_grad_(True)
    hl, rd, v = dev(h_local), dev(rays_d), dev(vf)
    out = _kernel_chain(fused, hl, torch.tensor(z).cuda(), rd, v, R, P, S)
    pk = [p_ for n_, p_ in awp.named_parameters() if not n_.startswith(("sample_feature_embed_layer", "MAM.conv."))]
    grads = torch.autograd.grad((out * torch.tensor(proj).cuda()).sum(), [hl, rd, v] + pk)
    c64 = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
    hl64, rd64, v64 = c64(h_local), c64(rays_d), c64(vf)
    out64 = ref.forward_from_local(hl64, torch.tensor(z, dtype=torch.float64), rd64, v64)
    names = [n_ for n_, _ in ref.named_parameters() if not n_.startswith(("sample_feature_embed_layer", "MAM.conv."))]
    pr = dict(ref.named_parameters())
    grads64 = torch.autograd.grad((out64 * torch.tensor(proj, dtype=torch.float64)).sum(), [hl64, rd64, v64] + [pr[n_] for n_ in names])
    assert (out.detach().cpu().double() - out64.detach()).abs().max().item() < 2e-6
    scale = max(float(g_.norm()) for g_ in grads64)
    for key, a, b in zip(["h_local", "rays_d", "view_feature"] + names, grads, grads64):
        if float(b.norm()) < 1e-6 * scale:            # analytically (near) zero: MAM.linear.bias behind the training-mode BatchNorm
            assert float(a.norm()) < 1e-4 * scale, key
            continue
        assert rel(a, b) < 1e-4, (variant, key, rel(a, b))
    if variant == "eval_mode":                        # the estimates are read, not written
        assert torch.equal(awp.MAM.Corr.convd[1].running_mean.cpu().double(), ref.MAM.Corr.convd[1].running_mean)


@pytest.mark.parametrize("R,P,S", [(40, 10, 128), (7, 5, 33), (3, 16, 17)])
def test_local_consumers_one_launch_backward_equals_the_two_kernels(R, P, S):
    """evd_awp_local_consumers_backward (the integration's and the MAM per-sample part's backwards in one pass over h_local) against
    evd_awp_feature_integration_bwd + evd_mam_local_backward: d h_local, d z_vals, d rays_d, d u within float32 rounding of the sums"""
    from evdeblurnerf_amd.awp import _LocalConsumers
    rs = np.random.RandomState(R + S)
    res = {}
    for mode in ("separate", "fused"):
        os.environ["EVD_AWP_LOCAL_BWD"] = mode
        try:
            hl = torch.tensor(np.abs(rs.standard_normal((R * P, S, 64))).astype(np.float32) * (0.6 if mode else 1)).cuda().requires_grad_(True) if mode == "separate" else res["in"][0].detach().clone().requires_grad_(True)
            if mode == "separate":
                z = torch.tensor(np.sort(rs.uniform(0, 1, (R * P, S)).astype(np.float32), -1)).cuda().requires_grad_(True)
                d = torch.tensor(rs.standard_normal((R * P, 3)).astype(np.float32)).cuda().requires_grad_(True)
                u = torch.tensor(rs.standard_normal(64).astype(np.float32)).cuda().requires_grad_(True)
                pr = [torch.tensor(rs.standard_normal(sh).astype(np.float32)).cuda() for sh in ((R * P, 64), (R, P, 64), (R, S, 64))]
                res["in"] = (hl, z, d, u, pr)
            else:
                _, z, d, u, pr = res["in"]
                z, d, u = (t.detach().clone().requires_grad_(True) for t in (z, d, u))
            h, hi, hs = _LocalConsumers.apply(hl, z, d, u, R, P, S)
            res[mode] = torch.autograd.grad((h * pr[0]).sum() + (hi * pr[1]).sum() + (hs * pr[2]).sum(), [hl, z, d, u])
        finally:
            os.environ.pop("EVD_AWP_LOCAL_BWD", None)
    for name, a, b in zip(("d h_local", "d z_vals", "d rays_d", "d u"), res["fused"], res["separate"]):
        assert rel(a, b) < 5e-6, (name, rel(a, b))
