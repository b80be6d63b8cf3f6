"""The AWP consumer on the GPU: sample_feature_embed_layer fused into one MFMA kernel that reads the fine level's geo
fragments (evd_awp_embed_forward / _backward, reference networks/dpnerf/awp.py:36-37,98-100), chained with the feature_integration
scan (awp.py:49-77), against golden G21 (the reference's AdaptiveWeightProposal.forward and its torch.autograd gradients), the C
oracle, and float64 torch autograd with the kernel's own ReLU pattern."""
import numpy as np
import pytest
import torch

import os
import sys

from conftest import load_golden
from evdeblurnerf_amd import weights as W

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

# fragment slots of the embedding's store (csrc/awp_embed.h, namespace awpstore)
A_GEO, A_E0, A_D_E0, A_D_GEO, A_TILE_FRAGS = 0, 8, 24, 40, 48
AABB = ((-1.5, -1.5, -1.0), (1.5, 1.5, 1.0))


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def phi(kk):
    return 8 * ((kk & 7) >> 2) + 4 * (kk >> 3) + (kk & 3)


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double().reshape(-1), torch.as_tensor(b).double().reshape(-1)
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def adecode(store, nsamp, slot, nfrag, dtype):
    """fragments [slot, slot + nfrag) of every tile of an awp store -> [nsamp, 16 nfrag], channel 16 j + phi(kk)"""
    tiles = (store.numel() - 256) // (A_TILE_FRAGS * 1024)
    v = store[:tiles * A_TILE_FRAGS * 1024].view(tiles, A_TILE_FRAGS, 64, 16)[:, slot:slot + nfrag].contiguous().view(dtype).float()
    v = v.view(tiles, nfrag, 2, 32, 8)
    out = torch.zeros((tiles, 32, nfrag * 16), dtype=torch.float32, device=store.device)
    for h in range(2):
        for e in range(8):
            out[:, :, torch.arange(nfrag) * 16 + phi(8 * h + e)] = v[:, :, h, :, e].permute(0, 2, 1)
    return out.reshape(tiles * 32, nfrag * 16)[:nsamp]


def _embed(prec="f16", seed=211):
    from evdeblurnerf_amd.awp import SampleFeatureEmbed
    sd = W.make_awp_embed_state_dict(seed)
    ws = [sd[f"sample_feature_embed_layer.{l}.weight"] for l in range(4)]
    bs = [sd[f"sample_feature_embed_layer.{l}.bias"] for l in range(4)]
    emb = SampleFeatureEmbed(ws, bs, precision=prec)
    flat = torch.cat([torch.tensor(t).reshape(-1) for l in range(4) for t in (ws[l], bs[l])]).cuda().requires_grad_(True)
    return emb, flat, ws, bs


def _mlp64(x, ws, bs, masks=None):
    """awp.py:98-100 in float64 torch; masks: the kernel's ReLU patterns (a 0/1 tensor per layer) instead of the own ones"""
    h = x
    for l in range(4):
        pre = h @ ws[l].t() + bs[l]
        h = pre * masks[l] if masks is not None else torch.relu(pre)
    return h


def _scan64(f, z, d):
    """awp.py:58-75 AS WRITTEN (zeros appended, cumprod along the channel axis of the previous sample's row), float64"""
    dists = (z[..., 1:] - z[..., :-1]) * torch.norm(d[..., None, :], dim=-1)
    alpha = -torch.exp(-f[..., :-1, :] * dists[..., None]) + 1
    alpha = torch.cat([alpha, torch.zeros_like(alpha[:, 0:1])], dim=-2)
    wts = alpha * torch.cumprod(torch.cat([torch.ones((alpha.shape[0], 1, alpha.shape[-1]), dtype=f.dtype, device=f.device), -alpha + (1. + 1e-10)], -2), -1)[:, :-1, :]
    return torch.sum(wts * f, dim=-2)


@pytest.mark.parametrize("prec,tol", [("f16", 3e-3), ("bf16", 3e-2)])
def test_sample_embed_forward_vs_reference_golden_and_oracle(O, prec, tol):
    """h_local of the fused kernel vs G21 (the reference module's own forward, captured by hooks) and, chained with the scan, the
    integrated features the reference hands to motion_feature_embed_layer; then a blurfactory-sized launch (2048 rays x 128 samples)
    against the oracle on a slice, ragged tail included."""
    from evdeblurnerf_amd.awp import feature_integration
    g = load_golden("G21_awp_sample_embed")
    emb, flat, ws, bs = _embed(prec)
    x = torch.tensor(g["depth_feature"], device="cuda")
    N, S, _ = x.shape
    with torch.no_grad():
        h_local = emb(flat, x).reshape(N, S, 64)
    err = (h_local.cpu() - torch.tensor(g["h_local"])).abs().max().item()
    scale = float(np.abs(g["h_local"]).max())
    print(f"[awp embed {prec}] h_local max abs err {err:.2e} (max |h_local| {scale:.2f})")
    assert err < tol * max(1.0, scale)
    P = g["proj"].shape[1]
    h = feature_integration(h_local.reshape(N // P, P, S, 64), torch.tensor(g["z"], device="cuda"), torch.tensor(g["rays_d"], device="cuda"))
    assert (h.cpu() - torch.tensor(g["h"][..., :64])).abs().max().item() < tol * max(1.0, float(np.abs(g["h"][..., :64]).max()))
    # oracle == golden on CPU is tests/test_oracle_golden.py; here the kernel vs the oracle at size
    rs = np.random.RandomState(5)
    n = 2048 * 128 - 37
    big = (rs.standard_normal((n, 128)) * 0.7).astype(np.float32)
    with torch.no_grad():
        hb = emb(flat, torch.tensor(big, device="cuda"))
    for lo in (0, n - 4096):
        ref = O.awp_sample_embed(big[lo:lo + 4096], ws, bs)
        assert np.abs(hb[lo:lo + 4096].cpu().numpy() - ref).max() < tol * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize("prec,tol", [("f16", 4e-3), ("bf16", 3e-2)])
@pytest.mark.parametrize("n", [4096, 1000])
def test_sample_embed_backward_matches_torch_autograd(prec, tol, n):
    """evd_awp_embed_backward (parameter gradients, d geo rows) vs float64 torch autograd of awp.py:98-100 with the kernel's own ReLU
    patterns (decoded from its store), over gradient magnitudes spanning several orders (the loss scale)."""
    emb, flat, ws, bs = _embed(prec)
    rs = np.random.RandomState(7)
    x = (rs.standard_normal((n, 128)) * 0.7).astype(np.float32)
    gout = (rs.standard_normal((n, 64)) * 1e-4 * np.exp(rs.uniform(-4, 0, (n, 1)))).astype(np.float32)
    xt = torch.tensor(x, device="cuda", requires_grad=True)
    h = emb(flat, xt)
    store = h.grad_fn.store
    (h * torch.tensor(gout, device="cuda")).sum().backward()
    dt = torch.float16 if prec == "f16" else torch.bfloat16
    masks = [(adecode(store, n, A_E0 + 4 * l, 4, dt) > 0).cpu().double() for l in range(4)]
    xq = adecode(store, n, A_GEO, 8, dt).cpu().double()           # the operand the kernel saw (rounded geo features)
    assert (xq - torch.tensor(x).double()).abs().max().item() < (2e-3 if prec == "f16" else 2e-2) * np.abs(x).max()
    w64 = [torch.tensor(w, dtype=torch.float64, requires_grad=True) for w in ws]
    b64 = [torch.tensor(b, dtype=torch.float64, requires_grad=True) for b in bs]
    x64 = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    r = _mlp64(x64, w64, b64, masks)
    assert (h.detach().cpu().double() - r).abs().max().item() < (5e-3 if prec == "f16" else 5e-2) * max(1.0, r.abs().max().item())
    (r * torch.tensor(gout, dtype=torch.float64)).sum().backward()
    got, errs = flat.grad.cpu().double(), {}
    for l, (wo, bo) in enumerate(emb.offsets):
        errs[f"w{l}"] = rel_l2(got[wo:wo + w64[l].numel()].reshape(w64[l].shape), w64[l].grad)
        errs[f"b{l}"] = rel_l2(got[bo:bo + 64], b64[l].grad)
    errs["geo"] = rel_l2(xt.grad.cpu(), x64.grad)
    print(f"[awp embed bwd {prec} n={n}] worst relative L2 {max(errs.values()):.2e}")
    assert max(errs.values()) < tol, {k: f"{v:.1e}" for k, v in errs.items()}


def test_embed_and_scan_gradients_against_the_reference_golden():
    """embed + feature_integration on the kernels, loss = <integrated features, proj>: gradients w.r.t. depth_feature and every
    sample_feature_embed_layer parameter vs torch.autograd ON THE REFERENCE MODULE (G21).  True ReLU patterns on both sides, so the
    bound is the float16 flips of near-zero pre-activations (measured and printed), not the arithmetic (previous test)."""
    from evdeblurnerf_amd.awp import feature_integration
    g = load_golden("G21_awp_sample_embed")
    emb, flat, ws, bs = _embed("f16")
    x = torch.tensor(g["depth_feature"], device="cuda", requires_grad=True)
    N, S, _ = x.shape
    P = g["proj"].shape[1]
    h_local = emb(flat, x).reshape(N // P, P, S, 64)
    h = feature_integration(h_local, torch.tensor(g["z"], device="cuda"), torch.tensor(g["rays_d"], device="cuda"))
    loss = (h * torch.tensor(g["proj"], device="cuda")).sum()
    assert abs(loss.item() - float(g["loss"])) < 2e-3 * max(1.0, abs(float(g["loss"])))
    loss.backward()
    got, errs = flat.grad.cpu(), {}
    for l, (wo, bo) in enumerate(emb.offsets):
        errs[f"w{l}"] = rel_l2(got[wo:wo + g[f"g.w{l}"].size].reshape(g[f"g.w{l}"].shape), g[f"g.w{l}"])
        errs[f"b{l}"] = rel_l2(got[bo:bo + 64], g[f"g.b{l}"])
    errs["depth_feature"] = rel_l2(x.grad.cpu(), g["g.depth_feature"])
    print("[awp embed + scan vs reference autograd]", {k: f"{v:.1e}" for k, v in errs.items()})
    assert max(errs.values()) < 3e-2, errs


def _fine_level(prec="f16"):
    from evdeblurnerf_amd.voxnerf import VoxelNeRFSampleFeatures
    nvox = 48 ** 3
    gsz = W.pdrf_grid_size(AABB[0], AABB[1], nvox)
    sd = W.make_pdrf_state_dict(71, gsz, input_ch=64 + 63, hidden_dim=256, geo_feat_dim=128, add_bias_color=True)
    net = VoxelNeRFSampleFeatures(sd, "", AABB, num_layers=2, hidden_dim=256, geo_feat_dim=128, num_layers_color=3, input_ch=64 + 63, app_dim=32,
                                  app_n_comp=(64, 16, 16), n_voxels=nvox, precision=prec)
    return net, sd


@pytest.mark.parametrize("prec", ["f16", "bf16"])
def test_fragment_path_equals_row_path_through_the_fine_level(prec):
    """The point of the fusion: the embedding reads the geo features as the fragments evd_voxel_mlp_train already stored, and hands
    their gradient back as fragments (evd_voxel_mlp_backward awp_store) -- no [R S, 128] float32 tensor in either direction.  Against
    the row path (feature rows out of the level, d_feature rows back in): h_local bit-identical (the rows are converted with the same
    rounding the fragments were), every gradient of the level within the half-precision rounding of the summed gradient."""
    from evdeblurnerf_amd.voxnerf import GeoFragments
    R, S = 70, 33
    rs = np.random.RandomState(11)
    pts = rs.uniform(-1, 1, (R, S, 3)).astype(np.float32)
    d = rs.normal(size=(R, 3))
    vd = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
    fts = (0.3 * rs.normal(size=(R, S, 64))).astype(np.float32)
    d_raw = torch.tensor((rs.normal(size=(R, S, 4)) * 1e-3).astype(np.float32), device="cuda")
    gh = torch.tensor((rs.normal(size=(R * S, 64)) * 1e-3).astype(np.float32), device="cuda")
    res = {}
    for path in ("rows", "fragments"):
        net, sd = _fine_level(prec)
        emb, eflat, _, _ = _embed(prec)
        flat = net.flat_params(sd)
        ft_t = torch.tensor(fts, device="cuda", requires_grad=True)
        pts_t, vd_t = torch.tensor(pts, device="cuda"), torch.tensor(vd, device="cuda")
        if path == "rows":
            raw, feat = net.mlp_train(flat, pts_t, vd_t, ft_t, want_feature=True)
            h_local = emb(eflat, feat)
        else:
            geo = GeoFragments()
            raw, geo.token = net.mlp_train(flat, pts_t, vd_t, ft_t, want_feature=geo)
            h_local = emb(eflat, geo)
        ((raw * d_raw).sum() + (h_local * gh).sum()).backward()
        res[path] = (h_local.detach().clone(), flat.grad.clone(), ft_t.grad.clone(), eflat.grad.clone())
    assert torch.equal(res["rows"][0], res["fragments"][0])
    tol = 3e-3 if prec == "f16" else 3e-2
    errs = {"level parameters": rel_l2(res["fragments"][1], res["rows"][1]), "d fts": rel_l2(res["fragments"][2], res["rows"][2]),
            "embed parameters": rel_l2(res["fragments"][3], res["rows"][3])}
    print(f"[awp fragments vs rows {prec}]", {k: f"{v:.1e}" for k, v in errs.items()})
    assert max(errs.values()) < tol, errs


from awp_standin import RefLikeAWP as _RefLikeAWP  # noqa: E402  (tools/awp_standin.py: the reference module's surface for the GPU box)


def test_fused_awp_module_in_the_training_forward():
    """NeRFAll.forward_train with awpnet = FusedAWP(module): `depth_feature` never exists as a tensor (the GeoFragments handle travels
    instead), rgb_awp and the gradients of EVERY parameter (the wrapped module's included) agree with the same model run with the
    plain torch module on the float32 feature rows; one optimizer step on the wrapped module's own parameters is seen by the next
    forward (re-pack)."""
    from test_gpu_train import _ToyRigidKernel, _c2f_model
    from evdeblurnerf_amd.awp import FusedAWP
    from evdeblurnerf_amd.voxnerf import GeoFragments
    torch.manual_seed(3)
    Kmat = W.synthetic_camera()
    R = 48
    rays = torch.tensor(W.synthetic_rays(3, R), device="cuda")
    target = torch.rand((R, 3), device="cuda")
    ref = _RefLikeAWP().cuda()
    kw = dict(force_naive=False, N_samples=16, N_importance=16, perturb=0.)
    out = {}
    for path in ("torch", "fused"):
        model, sd = _c2f_model("f16", 16)
        model.train()
        kern = _ToyRigidKernel().cuda()
        model.kernelsnet, model.kernel_type, model.use_awp = kern, "RBK", True
        model.awpnet = FusedAWP(ref) if path == "fused" else ref
        pc, pf = model.trainable_parameters(sd)
        seen = {}
        if path == "fused":
            orig = model.awpnet.forward
            model.awpnet.forward = lambda df, *a: (seen.__setitem__("df", df), orig(df, *a))[1]
        for p in ref.parameters():
            p.grad = None
        rgb, rgb0, other, tens = model.forward_train(400, 400, Kmat, rays, pc, pf, **kw)
        if path == "fused":
            assert isinstance(seen["df"], GeoFragments)
        loss = ((tens["rgb_awp"] - target) ** 2).mean() + 0.5 * ((rgb - target) ** 2).mean()
        loss.backward()
        out[path] = (tens["rgb_awp"].detach().clone(), {n: p.grad.detach().clone() for n, p in ref.named_parameters()},
                     pf["net"].grad.clone(), [g.grad.clone() for g in pf["grids"]])
    assert (out["fused"][0] - out["torch"][0]).abs().max().item() < 2e-3
    errs = {n: rel_l2(out["fused"][1][n], out["torch"][1][n]) for n in out["torch"][1]}
    errs["fine level"] = rel_l2(out["fused"][2], out["torch"][2])
    errs["fine planes"] = max(rel_l2(a, b) for a, b in zip(out["fused"][3][:3], out["torch"][3][:3]))
    print("[FusedAWP vs torch module]", {k: f"{v:.1e}" for k, v in errs.items()})
    assert max(errs.values()) < 5e-2, errs        # float16 embedding vs float32 torch layers: rounding + ReLU flips
    # an optimizer step on the wrapped module's parameters reaches the library's streams
    fused = FusedAWP(ref)
    x = torch.randn((5 * 7, 16, 128), device="cuda")
    z = torch.sort(torch.rand((35, 16), device="cuda"), -1)[0]
    rd, vf = torch.randn((35, 3), device="cuda"), torch.randn((7, 4), device="cuda")
    w0 = fused(x, z, rd, vf)
    with torch.no_grad():
        for p in ref.sample_feature_embed_layer.parameters():
            p.mul_(1.5)
    w1, w1_ref = fused(x, z, rd, vf), ref(x, z, rd, vf)
    assert (w1 - w0).abs().max().item() > 1e-4 and (w1 - w1_ref).abs().max().item() < 2e-3


# ------------------------------------------------------------------------------------------------ the MAM's per-sample part
from awp_standin import MAMLike as _MAMLike  # noqa: E402  (pinned to the reference's module by tests/test_oracle_golden.py::test_G22_mam)


def _t(a):
    return torch.as_tensor(np.ascontiguousarray(a), device="cuda")


_rel = rel_l2


def _mam_inputs(seed, R, P, S):
    rs = np.random.RandomState(seed)
    x_local = np.maximum(rs.standard_normal((R * P, S, 64)), 0).astype(np.float32)
    Wl = (rs.standard_normal((32, 64)) * 0.125).astype(np.float32)
    bl = (rs.standard_normal((32,)) * 0.1).astype(np.float32)
    v = (rs.standard_normal((1, 32, 1, 1)) * 0.5).astype(np.float32)
    return x_local, Wl, bl, v


@pytest.mark.parametrize("R,P,S", [(7, 10, 128), (3, 5, 64), (2, 1, 33), (2, 16, 200)])
def test_mam_local_equals_the_oracle(O, R, P, S):
    """evd_mam_local_forward + the caller's 64 -> 32 map against the oracle's as-written linear / logit / softmax / sums (mam.py:72-74,
    29-33); the kept softmax weights sum to 1 along their axis.  Float32: 2e-5 of the largest value."""
    from evdeblurnerf_amd.awp import mam_local
    x_local, Wl, bl, v = _mam_inputs(31 + P, R, P, S)
    hi, hs = mam_local(_t(x_local), _t(Wl), _t(v), R, P, S)
    inter = torch.nn.functional.linear(hi, _t(Wl), _t(bl)).transpose(1, 2).cpu().numpy()
    intra = torch.nn.functional.linear(hs, _t(Wl), _t(bl)).transpose(1, 2).cpu().numpy()
    o_inter, o_intra = O.mam_local(x_local, Wl, bl, v, P)
    assert np.abs(inter - o_inter).max() < 2e-5 * max(1.0, np.abs(o_inter).max())
    assert np.abs(intra - o_intra).max() < 2e-5 * max(1.0, np.abs(o_intra).max())


def test_mam_local_against_the_reference_golden():
    """G22 (the reference's MotionAggregationModule, run by tools/gen_golden.py): the kernel's sums through the module's own linear
    against the tensors the real module handed to Corr.conva / Corr.convb, and FusedAWP's restatement of the whole MAM (per-sample part
    on the library, the rest on the module's layers) against its output and autograd gradients."""
    from evdeblurnerf_amd.awp import mam_local, FusedAWP
    g = load_golden("G22_mam")
    R, P = g["x_global"].shape[:2]
    S = g["x_local"].shape[1]
    Wl, bl, v = _t(g["sd.linear.weight"]), _t(g["sd.linear.bias"]), _t(g["sd.Corr.line_conv_att.weight"])
    hi, hs = mam_local(_t(g["x_local"]), Wl, v, R, P, S)
    inter = torch.nn.functional.linear(hi, Wl, bl).transpose(1, 2).cpu().numpy()
    intra = torch.nn.functional.linear(hs, Wl, bl).transpose(1, 2).cpu().numpy()
    assert np.abs(inter - g["inter"]).max() < 2e-5 * max(1.0, np.abs(g["inter"]).max())
    assert np.abs(intra - g["intra"]).max() < 2e-5 * max(1.0, np.abs(g["intra"]).max())
    mam = _MAMLike(32, P - 1).train()
    mam.load_state_dict({k[3:]: torch.tensor(g[k]) for k in g if k.startswith("sd.")}, strict=True)
    holder = _RefLikeAWP(P=P, mam="corr").cuda()
    holder.MAM = mam.cuda()
    fused = FusedAWP(holder, "f16")
    xg, xl = _t(g["x_global"]).requires_grad_(True), _t(g["x_local"]).requires_grad_(True)
    out = fused._mam(xg, xl, R, P, S)
    assert np.abs(out.detach().cpu().numpy() - g["out"]).max() < 5e-5
    ps = [mam.linear.weight, mam.Corr.line_conv_att.weight]
    grads = torch.autograd.grad((out * _t(g["proj"])).sum(), [xg, xl] + ps)
    for got, key in zip(grads, ["g.x_global", "g.x_local", "g.linear.weight", "g.line_conv_att.weight"]):
        assert _rel(got.cpu().numpy(), g[key]) < 5e-5, key


@pytest.mark.parametrize("R,P,S", [(5, 10, 128), (3, 4, 40)])
def test_mam_local_backward_equals_float64_autograd(R, P, S):
    """evd_mam_local_backward against torch.autograd of the as-written float64 formula (linear, logit, two softmaxes, two sums) for
    d x_local, d linear.weight and d line_conv_att.weight.  Float32 kernels: 2e-5 relative L2."""
    from evdeblurnerf_amd.awp import mam_local
    x_local, Wl, bl, v = _mam_inputs(57 + P, R, P, S)
    rs = np.random.RandomState(5)
    pi, ps_ = rs.standard_normal((R, 32, P)), rs.standard_normal((R, 32, S))
    x64 = torch.tensor(x_local, dtype=torch.float64, requires_grad=True)
    W64 = torch.tensor(Wl, dtype=torch.float64, requires_grad=True)
    b64 = torch.tensor(bl, dtype=torch.float64)
    v64 = torch.tensor(v.reshape(-1), dtype=torch.float64, requires_grad=True)
    cur = (x64.reshape(R, P, S, 64) @ W64.t() + b64).permute(0, 3, 1, 2)                 # [R, 32, P, S]
    att = (cur * v64[None, :, None, None]).sum(1, keepdim=True)
    l64 = ((cur * att.softmax(-1)).sum(-1) * torch.tensor(pi)).sum() + ((cur * att.softmax(-2)).sum(-2) * torch.tensor(ps_)).sum()
    want = torch.autograd.grad(l64, [x64, W64, v64])
    xg, Wg, vg = _t(x_local).requires_grad_(True), _t(Wl).requires_grad_(True), _t(v).requires_grad_(True)
    hi, hs = mam_local(xg, Wg, vg, R, P, S)
    inter = torch.nn.functional.linear(hi, Wg, _t(bl)).transpose(1, 2)
    intra = torch.nn.functional.linear(hs, Wg, _t(bl)).transpose(1, 2)
    got = torch.autograd.grad((inter * _t(pi.astype(np.float32))).sum() + (intra * _t(ps_.astype(np.float32))).sum(), [xg, Wg, vg])
    for a, b, name in zip(got, want, ("x_local", "linear.weight", "line_conv_att.weight")):
        assert _rel(a.cpu().numpy().reshape(-1), b.numpy().reshape(-1)) < 2e-5, name


def test_fused_awp_with_the_reference_mam_structure_equals_plain_torch():
    """FusedAWP around a module with the reference's MotionAggregationModule structure (MAMLike): output and parameter gradients
    against the same module run in plain float32 torch on the float32 depth_feature tensor.  Float16 embedding against float32 torch
    layers (rounding + ReLU flips): 5e-2 of the whole gradient, 15 % per tensor (64 rays: single flips show in the small biases)."""
    from evdeblurnerf_amd.awp import FusedAWP
    torch.manual_seed(3)
    P, R, S = 5, 64, 64
    ref = _RefLikeAWP(P=P, mam="corr").cuda()
    rs = np.random.RandomState(9)
    df = _t((rs.standard_normal((R * P, S, 128)) * 0.5).astype(np.float32))
    z = _t(np.sort(rs.uniform(0, 1, (R * P, S)).astype(np.float32), -1))
    d = _t(rs.standard_normal((R * P, 3)).astype(np.float32))
    vf = _t(rs.standard_normal((R, 4)).astype(np.float32))
    proj = _t(rs.standard_normal((R, P)).astype(np.float32))
    names = [n_ for n_, _ in ref.named_parameters()]
    want_out = ref(df, z, d, vf)
    want = torch.autograd.grad((want_out * proj).sum(), list(ref.parameters()), allow_unused=True)
    fused = FusedAWP(ref, "f16")
    got_out = fused(df, z, d, vf)
    got = torch.autograd.grad((got_out * proj).sum(), list(ref.parameters()), allow_unused=True)
    assert np.abs((got_out - want_out).detach().cpu().numpy()).max() < 2e-3
    # MAM.linear.bias has an analytically zero gradient (a constant added to every curve is removed by the training-mode BatchNorm of
    # Corr.convd): both sides hold rounding noise there
    errs = {n_: _rel(a.cpu().numpy(), b.cpu().numpy()) for n_, a, b in zip(names, got, want)
            if b is not None and float(b.abs().max()) > 1e-6 and n_ != "MAM.linear.bias"}
    ga = torch.cat([a.reshape(-1) for a, b in zip(got, want) if b is not None])
    gb = torch.cat([b.reshape(-1) for b in want if b is not None])
    print("[FusedAWP, reference MAM structure, vs torch]", f"all {_rel(ga, gb):.1e}", {k: f"{v:.1e}" for k, v in errs.items()})
    # the MAM's own path is pinned to 5e-5 by test_mam_local_against_the_reference_golden; this is the wiring, through a float16 embedding
    assert _rel(ga, gb) < 5e-2 and max(errs.values()) < 0.15, errs


def test_fused_awp_per_ray_tail_as_a_captured_graph():
    """FusedAWP(graph_per_ray=True): the per-ray remainder (awp.py:105-117, mam.py:35-53; ~100 small launches forward, twice that
    backward) replayed as one hipGraph each way.  Same module, same inputs: output and the gradients of every parameter equal the
    eager path's (the captured kernels ARE the eager kernels), over two steps with an optimizer update in between."""
    from evdeblurnerf_amd.awp import FusedAWP
    torch.manual_seed(5)
    rs = np.random.RandomState(17)
    R, P, S = 64, 5, 32
    ref = _RefLikeAWP(P=P, mam="corr").cuda()
    ref2 = _RefLikeAWP(P=P, mam="corr").cuda()
    ref2.load_state_dict(ref.state_dict())
    # (the same torch remainder, eager and captured.  Against the KERNEL remainder this loop differs by 7e-3 at its second step: one
    # element of the 10 240 of z = x_global + BatchNorm(...) lies within rounding of the leaky ReLU's kink there, the two float32 paths put it
    # on different sides, and one element's whole gradient is 1e-2 of the norm at this batch size -- float64 sides with the kernels)
    eager, graphed = FusedAWP(ref, "f16", tail_kernels=False), FusedAWP(ref2, "f16", graph_per_ray=True)
    opts = [torch.optim.SGD(m.parameters(), lr=1e-2) for m in (ref, ref2)]
    for step in range(3):
        df = _t((0.5 * rs.standard_normal((R * P, S, 128))).astype(np.float32))
        z = _t(np.sort(rs.uniform(0, 1, (R * P, S)).astype(np.float32), -1))
        d = _t(rs.standard_normal((R * P, 3)).astype(np.float32))
        vf = _t(rs.standard_normal((R, 4)).astype(np.float32))
        proj = _t(rs.standard_normal((R, P)).astype(np.float32))
        outs, grads = [], []
        for fused, m, opt in ((eager, ref, opts[0]), (graphed, ref2, opts[1])):
            opt.zero_grad(set_to_none=True)
            df_ = df.clone().requires_grad_(True)
            out = fused(df_, z, d.clone().requires_grad_(True), vf)
            (out * proj).sum().backward()
            outs.append(out.detach().clone())
            grads.append({n_: (p.grad.detach().clone() if p.grad is not None else None) for n_, p in m.named_parameters()})
            grads[-1]["d depth_feature"] = df_.grad.detach().clone()
            opt.step()
        # the BatchNorm statistics saw the same real batches on both sides: the capture's warm-up passes on random data were undone
        for (ka, va), (kb, vb) in zip(sorted(ref.state_dict().items()), sorted(ref2.state_dict().items())):
            if "running_" in ka or "num_batches_tracked" in ka:
                assert torch.allclose(va.float(), vb.float(), rtol=1e-3, atol=1e-5), (step, ka, va, vb)
        ref2.load_state_dict(ref.state_dict())      # (in place: the captured graph keeps reading these tensors) both sides start every step equal --
        # left alone, the two float16 embeddings' rounding-level differences grow through the ReLU flips of later steps
        assert graphed._graphed, "the graph was not built"
        assert float((outs[0] - outs[1]).abs().max()) < 1e-5, step
        for n_ in grads[0]:
            a, b = grads[0][n_], grads[1][n_]
            if a is None or b is None:
                assert (a is None or float(a.abs().max()) == 0.0) and (b is None or float(b.abs().max()) == 0.0), n_
                continue
            if n_ == "MAM.linear.bias":      # analytically zero (a constant added to every curve is removed by the training-mode BatchNorm): rounding noise
                assert float(a.abs().max()) < 1e-4 and float(b.abs().max()) < 1e-4
                continue
            # (two models, each with float16 fragments + float atomics in its own embedding: rounding-level differences)
            assert float((a - b).abs().max()) <= 2e-3 * (float(a.abs().max()) + 1e-6), (step, n_, float((a - b).abs().max()), float(a.abs().max()))
