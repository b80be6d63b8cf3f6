"""GPU parity at the REAL workload sizes: the 4096 x 128 metric render of every arithmetic mode against the
CPU oracle (not against another kernel of this library), on seed-derived AND on trained weights; the shipped blurfactory
configuration (PDRF grids 293x293x195 / 586x586x390) at 4096 rays x (64 + 64) and as one 400 x 400 frame at 64 + 128, against
the oracle on a ray slice plus size-independent properties over all rays.  Tolerances are stated next to each assert."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, maxabs, z_mismatch
from evdeblurnerf_amd import weights as W

pytestmark = pytest.mark.gpu

DEV = "cuda"
NERF_KW = dict(ndc=True, near=0., far=1., use_viewdirs=True, N_importance=0, retraw=False)


def T(x):
    return torch.as_tensor(np.ascontiguousarray(x), device=DEV)


def N(x):
    return x.detach().cpu().numpy()


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def _nerf_args():
    from types import SimpleNamespace
    return SimpleNamespace(mode="nerf", netdepth=8, netwidth=256, multires=10, multires_views=4, use_viewdirs=True,
                           rgb_activate="sigmoid", sigma_activate="relu", N_importance=0)


def _render_modes(sd, rays, S=128):
    from evdeblurnerf_amd.renderer import NeRFAll
    K = W.synthetic_camera()
    return {p: N(NeRFAll(_nerf_args(), sd, precision=p).eval().render(400, 400, K, rays=rays, N_samples=S, **NERF_KW)[0])
            for p in ("f32", "f16x3", "f16", "bf16")}


def test_metric_render_every_mode_vs_oracle_at_full_size(O):
    """4096 rays x 128 samples through the 8x256 network: RGB L-inf of each mode vs oracle/evd_oracle.c on the bench's rays and
    (seed-derived) weights.  f32 / f16x3 / f16 <= 1e-4 (north_star's bound); bf16 (2^-8 operands) <= 3e-2, reported."""
    sd = W.prefixed(W.make_nerf_state_dict(21), "mlp_coarse")
    rays = W.synthetic_rays(100, 4096)
    ref = O.render_nerf(O.Nerf(sd, "mlp_coarse."), None, O.make_cfg(N_samples=128), rays)["rgb"]
    out = _render_modes(sd, T(rays))
    err = {p: maxabs(v, ref) for p, v in out.items()}
    print("RGB L-inf vs the oracle, 4096 x 128, seed-derived weights:", {k: f"{v:.2e}" for k, v in err.items()})
    assert err["f32"] < 1e-4 and err["f16x3"] < 1e-4 and err["f16"] < 1e-4
    assert err["bf16"] < 3e-2


@pytest.fixture(scope="module")
def trained():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import trained_weights as TW
    sd, rep = TW.train_nerf(iters=1000)
    print("trained weights:", rep)
    return sd, rep


def test_trained_weights_every_mode_vs_oracle(O, trained):
    """The same comparison on TRAINED weights (tools/trained_weights.py: 1000 Adam iterations of the library's own training path on
    an analytic scene; densities of ~10 instead of ~1).  The float32-grade modes must hold 1e-4 whatever the weights;
    the single-product float16 / bfloat16 modes are reported and bounded loosely (their error grows with the densities)."""
    sd, rep = trained
    assert rep["mse_last"] < 0.5 * rep["mse_first"]              # it did train
    assert rep["sigma_max"] > 5.0                                # and it is not the near-constant initial field any more
    rays = W.synthetic_rays(100, 4096)
    ref = O.render_nerf(O.Nerf(sd, "mlp_coarse."), None, O.make_cfg(N_samples=128), rays)["rgb"]
    out = _render_modes(sd, T(rays))
    err = {p: maxabs(v, ref) for p, v in out.items()}
    print("RGB L-inf vs the oracle, 4096 x 128, trained weights:", {k: f"{v:.2e}" for k, v in err.items()})
    assert err["f32"] < 1e-4 and err["f16x3"] < 1e-4
    assert err["f16"] < 5e-3 and err["bf16"] < 1e-1 and err["f16"] < err["bf16"]


# ----------------------------------------------------------------------------------------------- blurfactory configuration

@pytest.fixture(scope="module")
def blurfactory(O):
    """both PDRF levels at the shipped grid sizes (41 M grid values, 165 MB) + the oracle's view of the same parameters"""
    sd = W.make_blurfactory_state_dict(31, sigma_gain=3.0)
    lo, hi = W.BLURFACTORY_AABB
    gc, gf = W.pdrf_grid_size(lo, hi, W.BLURFACTORY_COARSE_VOXELS), W.pdrf_grid_size(lo, hi, W.BLURFACTORY_FINE_VOXELS)
    assert gc == [293, 293, 195] and gf == [586, 586, 390]
    vc = O.Voxel(sd, "mlp_coarse.", gc, lo + hi, input_ch=95)
    vf = O.Voxel(sd, "mlp_fine.", gf, lo + hi, input_ch=127, hidden_dim=256, geo_feat_dim=128, rgb_act="none")
    return sd, vc, vf


def _c2f_checks(O, model, vc, vf, rays, Ni, prec, tol, n_oracle=256):
    K = W.synthetic_camera()
    kw = dict(ndc=True, near=0., far=1., use_viewdirs=True, N_samples=64, N_importance=Ni, retraw=True, perturb=0., raw_noise_std=0.)
    R = rays.shape[0]
    rgb, depth, acc, ex = model.render(400, 400, K, rays=T(rays), **kw)
    # ---- properties over ALL rays (index arithmetic at 586x586x390, float16 grid copies, the merge)
    assert torch.isfinite(rgb).all() and torch.isfinite(depth).all()
    assert np.allclose(N(acc), 1.0, atol=3e-5)                                     # last alpha is forced to 1 (voxnerf.py:188-189)
    w = N(ex["weights"])
    assert (w >= -1e-6).all() and np.allclose(w.sum(-1), 1.0, atol=3e-5)
    zv = N(ex["z_vals"])
    assert zv.shape == (R, 64 + Ni) and (np.diff(zv, axis=-1) >= 0).all()          # merged samples are sorted
    assert np.array_equal(N(ex["z_vals0"]), np.broadcast_to(N(ex["z_vals0"])[:1], (R, 64)))     # perturb = 0: one z ladder
    assert float(rgb.std()) > 1e-3                                                 # the field is not constant (sigma_gain)
    # ---- against the oracle on a slice spread over the batch
    idx = np.linspace(0, R - 1, n_oracle).astype(np.int64)
    ref = O.render_c2f(vc, vf, O.make_cfg(N_samples=64, N_importance=Ni), rays[idx])
    e, e0 = maxabs(N(rgb)[idx], ref["rgb"]), maxabs(N(ex["rgb0"])[idx], ref["rgb0"])
    frac, worst = z_mismatch(zv[idx], ref["z_vals"], tol={"bf16": 5e-3, "f16": 5e-4}.get(prec, 5e-5))
    print(f"[blurfactory {prec} 64+{Ni}, {R} rays] RGB L-inf vs oracle ({n_oracle} rays): fine {e:.3e} coarse {e0:.3e}; moved samples {frac:.4f}")
    assert e < tol and e0 < tol
    assert maxabs(N(ex["weights0"])[idx], ref["weights0"]) < tol
    assert frac < {"bf16": 0.3, "f16": 0.05}.get(prec, 0.01) and worst < 1.0 / 63 + 1e-3
    return rgb, kw


@pytest.mark.parametrize("prec,tol", [("f32", 1e-4), ("f16x3", 1e-4), ("f16c", 1e-4), ("f16", 3e-4), ("bf16", 3e-2)])
def test_blurfactory_event_batch_at_real_grid_sizes(prec, tol, O, blurfactory):
    """BASELINE configs 2/3: 4096 event rays x (64 + 64) samples through both PDRF levels at grids 293x293x195 / 586x586x390."""
    from evdeblurnerf_amd.renderer import NeRFAll
    sd, vc, vf = blurfactory
    model = NeRFAll(W.blurfactory_args(64), sd, precision=prec).eval()
    assert model.mlp_coarse.gridSize == [293, 293, 195] and model.mlp_fine.gridSize == [586, 586, 390]
    rays = W.synthetic_rays(5, 4096)
    rgb, kw = _c2f_checks(O, model, vc, vf, rays, 64, prec, tol)
    if prec == "f16x3":
        K = W.synthetic_camera()
        perm = torch.randperm(4096, device=DEV)
        assert torch.equal(model.render(400, 400, K, rays=T(rays)[perm], **kw)[0], rgb[perm])          # rays are independent
        assert torch.equal(model.render(400, 400, K, chunk=1000, rays=T(rays), **kw)[0], rgb)          # chunking changes nothing


@pytest.mark.parametrize("prec,tol", [("f16x3", 1e-4), ("f16c", 1e-4), ("f16", 3e-4)])
def test_blurfactory_full_frame_at_real_grid_sizes(prec, tol, O, blurfactory):
    """BASELINE config 5: one 400 x 400 view = 160 000 rays, 64 + 128 samples (render_kwargs_test), get_rays on the device."""
    from evdeblurnerf_amd.rays import get_rays
    from evdeblurnerf_amd.renderer import NeRFAll
    sd, vc, vf = blurfactory
    model = NeRFAll(W.blurfactory_args(128), sd, precision=prec).eval()
    K = W.synthetic_camera()
    c2w = W.synthetic_pose(40)[:3, :4].astype(np.float32)
    o, d = get_rays(400, 400, K, T(c2w))
    rays = N(torch.stack([o, d], -1).reshape(-1, 3, 2))
    rgb, kw = _c2f_checks(O, model, vc, vf, rays, 128, prec, tol)
    kw.pop("retraw")
    frames, _ = model.render_path(400, 400, K, 1 << 22, [T(c2w)], kw)
    assert frames.shape == (1, 400, 400, 3) and torch.equal(frames[0].reshape(-1, 3), rgb)              # render_path == render


# ----------------------------------------------------------------------------------------------- compensated float16 mode (f16c)

@pytest.fixture(scope="module")
def trained_10k():
    """tests/golden/trained/nerf_8x256_10k.npz: the 8x256 network after 10 000 Adam iterations of the library's own training path on the
    analytic scene (`python tools/trained_weights.py --iters 10000 --save ...` on the GPU box, 54 s; mse 0.13 -> 1.5e-5, densities up to
    24, hidden weights 1.1-1.3x their initial magnitude) -- VERDICT r2 asked for the margin of f16c on a longer-trained network."""
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "trained", "nerf_8x256_10k.npz")))


def test_f16c_holds_the_bound_on_seed_and_trained_weights(O, trained, trained_10k):
    """EVD_PREC_F16C (float16 MFMA + two block-scaled fp6 MFMA products of the operands' rounding residuals, mlp_pipe_c.h): RGB L-inf
    vs the oracle at 4096 x 128 <= 1e-4 on the seed-derived weights, on weights trained in this session (1000 iterations) AND on the
    committed 10 000-iteration network (where the single-product float16 mode is at 2.5e-4); measured 8e-7 / 1.8e-5 / 1.5e-5.
    Bound asserted with a 2x margin."""
    from evdeblurnerf_amd.renderer import NeRFAll
    K = W.synthetic_camera()
    rays = W.synthetic_rays(100, 4096)
    for name, sd in (("seed", W.prefixed(W.make_nerf_state_dict(21), "mlp_coarse")), ("trained", trained[0]), ("trained 10k", trained_10k)):
        ref = O.render_nerf(O.Nerf(sd, "mlp_coarse."), None, O.make_cfg(N_samples=128), rays)["rgb"]
        got = {p: N(NeRFAll(_nerf_args(), sd, precision=p).eval().render(400, 400, K, rays=T(rays), N_samples=128, **NERF_KW)[0]) for p in ("f16c", "f16")}
        err = {p: maxabs(v, ref) for p, v in got.items()}
        print(f"RGB L-inf vs the oracle, 4096 x 128, {name} weights:", {k: f"{v:.2e}" for k, v in err.items()})
        assert err["f16c"] < 5e-5
        assert err["f16c"] < 0.5 * err["f16"]


def test_f16c_ragged_sizes_goldens_and_rejections(O):
    """The compensated mode at sizes that are not multiples of its 128-sample workgroups (1 ray x 1 sample ... 37 x 65), against the
    exact-float32 kernel; under larger hidden weights (x1.4, the stress of test_precision_modes_under_larger_weights); with activations
    beyond the float16 range (finite outputs); after a device re-pack of new parameters (bit-identical to a fresh handle); and the
    configurations it is not built for are rejected, not silently served by another kernel."""
    from evdeblurnerf_amd import _lib as L
    from evdeblurnerf_amd.nerf import NeRF
    sd = W.make_nerf_state_dict(5)
    net = NeRF(sd)
    rs = np.random.RandomState(3)
    # (700 x 97 = 530.5 tiles: the persistent workgroups -- one per CU -- walk several tiles each and the last one is ragged)
    for R, S in ((1, 1), (1, 128), (3, 43), (37, 65), (129, 127), (700, 97), (257, 128), (1000, 66)):      # 257 tiles: one workgroup walks two
        rb = np.zeros((R, 11), np.float32)
        rb[:, :3] = rs.uniform(-1, 1, (R, 3)); rb[:, 3:6] = rs.uniform(-1, 1, (R, 3)); rb[:, 7] = 1
        vd = rs.standard_normal((R, 3)); rb[:, 8:11] = vd / np.linalg.norm(vd, axis=1, keepdims=True)
        z = np.sort(rs.uniform(0, 1, (R, S)).astype(np.float32), -1)
        ref = N(net.mlpforward(T(rb), T(z), precision="f32")[0])
        got = N(net.mlpforward(T(rb), T(z), precision="f16c")[0])
        assert got.shape == (R, S, 4) and maxabs(got, ref) < 2e-5, (R, S, maxabs(got, ref))
    # larger weights
    big = dict(sd)
    for k in list(big):
        if "pts_linears" in k and k.endswith("weight") and not k.endswith("pts_linears.0.weight"):
            big[k] = (big[k] * 1.4).astype(np.float32)
    nb = NeRF(big)
    rb = np.zeros((512, 11), np.float32)
    rb[:, :3] = rs.uniform(-1, 1, (512, 3)); rb[:, 3:6] = rs.uniform(-1, 1, (512, 3)); rb[:, 7] = 1; rb[:, 8:11] = [0, 0, -1]
    z = np.sort(rs.uniform(0, 1, (512, 64)).astype(np.float32), -1)
    ref = N(nb.mlpforward(T(rb), T(z), precision="f32")[0])
    e16 = maxabs(N(nb.mlpforward(T(rb), T(z), precision="f16")[0]), ref)
    e16c = maxabs(N(nb.mlpforward(T(rb), T(z), precision="f16c")[0]), ref)
    print(f"raw L-inf vs f32 with 1.4x hidden weights: f16 {e16:.2e}, f16c {e16c:.2e}; max |raw| {np.abs(ref).max():.2f}")
    assert e16c < 0.2 * e16
    # float16 range: finite
    hot = dict(sd); hot["pts_linears.0.weight"] = (sd["pts_linears.0.weight"] * 1e5).astype(np.float32)
    out = N(NeRF(hot).mlpforward(T(rb), T(z), precision="f16c")[0])
    assert np.isfinite(out).all()
    # device re-pack
    sd2 = W.make_nerf_state_dict(6)
    fresh = N(NeRF(sd2).mlpforward(T(rb), T(z), precision="f16c")[0])
    net2 = NeRF(sd)
    # the flat parameter tensor in the library's canonical order
    parts = []
    for l in range(8):
        parts += [sd2[f"pts_linears.{l}.weight"].ravel(), sd2[f"pts_linears.{l}.bias"].ravel()]
    for k in ("views_linears.0", "feature_linear", "alpha_linear", "rgb_linear"):
        parts += [sd2[k + ".weight"].ravel(), sd2[k + ".bias"].ravel()]
    flat = T(np.concatenate(parts).astype(np.float32))
    net2.load_params(flat)
    again = N(net2.mlpforward(T(rb), T(z), precision="f16c")[0])
    assert np.array_equal(again, fresh)
    # rejections
    with pytest.raises(L.EvdError):
        net.mlpforward(T(rb), T(z), want_feature=True, precision="f16c")
    with pytest.raises(L.EvdError):
        NeRF(W.make_nerf_state_dict(1, D=4, W=64, skips=(2,)), D=4, W=64, skips=(2,)).mlpforward(T(rb), T(z), precision="f16c")


@pytest.mark.parametrize("S,white,perturb", [(128, False, 0.0), (64, True, 1.0), (32, False, 1.0)])
def test_f16c_fused_step_matches_the_separate_kernels(S, white, perturb, monkeypatch):
    """mode='nerf' without importance samples in precision f16c can run the whole step in ONE launch (z stratification in the MLP kernel's
    prologue, raw2outputs in its epilogue: nerf_mlp_c_kernel.h FUSE).  Every output of render() -- rgb / depth / acc, and with retraw
    z_vals / weights -- must equal what the separate entries give on the same inputs (evd_sample_z -> evd_nerf_mlp ->
    evd_raw2outputs): z bit-exact, the composited values to 2e-6 (scan association)."""
    from evdeblurnerf_amd.renderer import NeRFAll
    from evdeblurnerf_amd import _lib as L
    import ctypes as C
    monkeypatch.setenv("EVD_FUSE_STEP", "1")        # opt-in: measured 0.5 % slower than the separate kernels at 4096 x 128 (evd_api.hip)
    sd = W.prefixed(W.make_nerf_state_dict(21), "mlp_coarse")
    K = W.synthetic_camera()
    R = 1000 if S == 128 else 777
    rays = T(W.synthetic_rays(9, R))
    model = NeRFAll(_nerf_args(), sd, precision="f16c").eval()
    torch.manual_seed(3)
    t_rand = torch.rand((R, S), device=DEV) if perturb else None
    kw = dict(ndc=True, near=0., far=1., use_viewdirs=True, N_samples=S, N_importance=0, white_bkgd=white, perturb=perturb, retraw=True)
    if t_rand is not None:
        kw["t_rand"] = t_rand
    rgb, depth, acc, extras = model.render(400, 400, K, rays=rays, **kw)
    # the separate entries
    net = model.mlp_coarse
    cfg = model._cfg(400, 400, float(K[0][0]), True, 0., 1., S, 0, False, perturb, white)
    rb = torch.empty((R, 11), device=DEV)
    z = torch.empty((R, S), device=DEV)
    L.check(L.lib().evd_ray_batch(C.byref(cfg), L.ptr(rays), R, L.ptr(rb), L.stream_ptr()))
    L.check(L.lib().evd_sample_z(C.byref(cfg), L.ptr(rb), 11, R, L.ptr(t_rand) if t_rand is not None else None, L.ptr(z), L.stream_ptr()))
    raw, _ = net.mlpforward(rb, z, precision="f16c")
    r_rgb, _, r_acc, r_w, r_depth, _ = net.raw2outputs(raw, z, rb[:, 3:6].contiguous(), white_bkgd=white)
    assert torch.equal(extras["z_vals"], z)
    for name, a, b in (("rgb", rgb, r_rgb), ("depth", depth, r_depth), ("acc", acc, r_acc), ("weights", extras["weights"], r_w)):
        d = maxabs(N(a), N(b))
        print(f"[fused step S={S}] {name}: {d:.2e}")
        assert d < 2e-6, name
