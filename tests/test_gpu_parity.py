"""GPU parity tests: the HIP path (through the C ABI, via the Python mirror) against the CPU oracle on the same
seeded inputs, against the committed reference goldens, and -- at BASELINE.json's full sizes -- through
size-independent properties. Tolerances are stated next to each assert; the float32-grade modes
(f32, f16x3) are held to the north-star bound of 1e-4 RGB L-inf, bf16 is reported and loosely bounded."""
import numpy as np
import pytest
import torch

from conftest import load_golden, maxabs, sample_pdf_flip_report, z_mismatch
from evdeblurnerf_amd import weights as W

pytestmark = pytest.mark.gpu

DEV = "cuda"


def T(x):
    return torch.as_tensor(np.ascontiguousarray(x), device=DEV)


def N(x):
    return x.detach().cpu().numpy()


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def test_library_loads_and_sees_gpu():
    from evdeblurnerf_amd import _lib as L
    assert L.lib().evd_device_count() >= 1


def test_embed_matches_golden_and_oracle(O):
    from evdeblurnerf_amd.embedding import get_embedder
    g = load_golden("G1_embedder")
    for Lf, key in ((10, "pe10"), (4, "pe4"), (2, "pe2")):
        e, dim = get_embedder(Lf)
        out = N(e(T(g["x"])))
        assert out.shape[-1] == dim
        assert maxabs(out, g[key]) < 2e-6          # ocml sinf/cosf vs SLEEF, args up to 40*512
        assert maxabs(out, O.embed(g["x"], Lf)) < 2e-6


def test_rays_match_golden(O):
    from evdeblurnerf_amd.rays import get_rays, get_rays_pix, get_ndc_rays
    g = load_golden("G6_rays")
    o, d = get_rays(60, 80, g["Kn"], torch.as_tensor(g["c2w"]))
    assert maxabs(N(o)[::7, ::5], g["rays_o_full"]) == 0.0
    assert maxabs(N(d)[::7, ::5], g["rays_d_full"]) < 1e-6
    K = W.synthetic_camera()
    op, dp = get_rays_pix(T(g["coords"]), K, T(g["poses"]))
    assert maxabs(N(op), g["rays_o_pix"]) == 0.0
    assert maxabs(N(dp), g["rays_d_pix"]) < 1e-6
    on, dn = get_ndc_rays(400, 400, float(K[0, 0]), 1.0, T(g["rays_o_pix"]), T(g["rays_d_pix"]))
    assert maxabs(N(on), g["ndc_o"]) < 2e-6
    assert maxabs(N(dn), g["ndc_d"]) < 2e-6
    # bit-exact against the oracle's unfused float32 arithmetic
    oo, od = O.ndc_rays(400, 400, float(K[0, 0]), 1.0, g["rays_o_pix"], g["rays_d_pix"])
    assert maxabs(N(on), oo) <= 1e-7 and maxabs(N(dn), od) <= 1e-7


def _nerf_pair(seed_c=11, seed_f=12, **kw):
    from evdeblurnerf_amd.nerf import NeRF
    return (NeRF(W.make_nerf_state_dict(seed_c), **kw), NeRF(W.make_nerf_state_dict(seed_f), **kw))


@pytest.mark.parametrize("S", [64, 128, 33])
def test_raw2outputs_matches_golden(S, O):
    from evdeblurnerf_amd.nerf import NeRF
    g = load_golden("G3_nerf_raw2outputs")
    raw, z, d = g[f"raw_S{S}"], g[f"z_S{S}"], g[f"d_S{S}"]
    sd = W.make_nerf_state_dict(1)
    cases = {"plain": ({}, {}), "white": ({}, dict(white_bkgd=True)), "rmnear": (dict(render_rmnearplane=20), {}),
             "relu_rgb": (dict(rgb_activate="relu"), {}), "none_rgb": (dict(rgb_activate="none"), {}),
             "softplus": (dict(sigma_activate="softplus"), {})}
    for tag, (ckw, call) in cases.items():
        net = NeRF(sd, **ckw)
        rgb, dens, acc, wts, depth, fmap = net.raw2outputs(T(raw), T(z), T(d), None, 0, **call)
        # wavefront product scan vs sequential cumprod: a few ulp on O(1) values
        assert maxabs(N(rgb), g[f"rgb_S{S}_{tag}"]) < 5e-6, tag
        assert maxabs(N(acc), g[f"acc_S{S}_{tag}"]) < 5e-6, tag
        assert maxabs(N(depth), g[f"depth_S{S}_{tag}"]) < 5e-6, tag
        assert maxabs(N(wts), g[f"weights_S{S}_{tag}"]) < 5e-6, tag
        if tag == "plain":
            assert maxabs(N(dens), g[f"density_S{S}"]) < 2e-6
    net = NeRF(sd)
    fmap = net.raw2outputs(T(raw), T(z), T(d), T(g[f"feat_S{S}"]), 0)[5]
    assert maxabs(N(fmap), g[f"fmap_S{S}"]) < 5e-6


@pytest.mark.parametrize("S", [64, 128])
def test_voxel_channel_order_raw2outputs(S):
    """sigma first / rgb after (voxnerf.py:172,179) through the same C entry."""
    from evdeblurnerf_amd import _lib as L
    g = load_golden("G4_voxel_raw2outputs")
    raw, z, d = T(g[f"raw_S{S}"]), T(g[f"z_S{S}"]), T(g[f"d_S{S}"])
    R = raw.shape[0]
    for tag, act in (("coarse", "relu"), ("fine", "none")):
        rgb = torch.empty((R, 3), device=DEV)
        acc = torch.empty((R,), device=DEV)
        depth = torch.empty((R,), device=DEV)
        wts = torch.empty((R, S), device=DEV)
        L.check(L.lib().evd_raw2outputs(L.ptr(raw), L.ptr(z), L.ptr(d), 3, R, S, 4, 0, 1, 3, L.ACT[act], L.ACT["relu"], 0, 0.0, None,
                                        L.ptr(rgb), None, L.ptr(acc), L.ptr(wts), L.ptr(depth), None, 0, None, L.stream_ptr()))
        assert maxabs(N(rgb), g[f"rgb_S{S}_{tag}"]) < 5e-6
        assert maxabs(N(wts), g[f"weights_S{S}_{tag}"]) < 5e-6
        assert maxabs(N(depth), g[f"depth_S{S}_{tag}"]) < 5e-6
    # the mirror's VoxelNeRFBase.raw2outputs (reference 5-tuple order) incl. its autograd node: d raw vs torch autograd
    from evdeblurnerf_amd.voxnerf import VoxelNeRFRayFeatures
    gc = W.pdrf_grid_size(AABB[0], AABB[1], 24 ** 3)
    vox = VoxelNeRFRayFeatures(W.make_pdrf_state_dict(5, gc, input_ch=95, hidden_dim=64, geo_feat_dim=15), "", AABB, n_voxels=24 ** 3)
    raw_a = raw.clone().requires_grad_(True)
    rgb, dens, acc, wts, depth = vox.raw2outputs(raw_a, z, d)
    assert maxabs(N(rgb), g[f"rgb_S{S}_coarse"]) < 5e-6 and maxabs(N(depth), g[f"depth_S{S}_coarse"]) < 5e-6
    gr = torch.randn_like(rgb)
    (rgb * gr).sum().backward()
    raw_b = raw.double().clone().requires_grad_(True)
    dists = (z[:, 1:] - z[:, :-1]).double() * d.double().norm(dim=-1, keepdim=True)
    alpha = torch.cat([1.0 - torch.exp(-torch.relu(raw_b[:, :-1, 0]) * dists), torch.ones_like(dists[:, :1])], -1)
    Tt = torch.cumprod(torch.cat([torch.ones_like(dists[:, :1]), 1.0 - alpha + 1e-10], -1), -1)[:, :-1]
    ((alpha * Tt)[..., None] * torch.relu(raw_b[..., 1:]) * gr.double()[:, None, :]).sum().backward()
    assert float((raw_a.grad.double() - raw_b.grad).norm() / raw_b.grad.norm()) < 2e-5
    raw16 = T(g[f"raw16_S{S}"])
    fm = torch.empty((R, 15), device=DEV)
    wts = torch.empty((R, S), device=DEV)
    L.check(L.lib().evd_raw2outputs(L.ptr(raw16), L.ptr(z), L.ptr(d), 3, R, S, 16, 0, 1, 15, L.ACT["relu"], L.ACT["relu"], 0, 0.0, None,
                                    L.ptr(fm), None, None, L.ptr(wts), None, None, 0, None, L.stream_ptr()))
    assert maxabs(N(fm), g[f"fmap16_S{S}"]) < 5e-6


@pytest.mark.parametrize("S,Ns", [(64, 64), (64, 128), (128, 64), (17, 9)])
def test_sample_pdf_merge(S, Ns, O):
    from evdeblurnerf_amd.rays import sample_pdf_merge
    g = load_golden("G5_sample_pdf")
    key = f"S{S}_N{Ns}"
    bins, w, u = g[f"bins_{key}"], g[f"w_{key}"], g[f"u_{key}"]
    R = bins.shape[0]
    # rebuild a z row whose mid-points are the golden bins (z[0] = bins[0] - small) and full-width weights
    z = np.zeros((R, S), np.float64)
    z[:, 0] = bins[:, 0].astype(np.float64) - 1e-3
    for i in range(S - 1):
        z[:, i + 1] = 2.0 * bins[:, i].astype(np.float64) - z[:, i]
    z = z.astype(np.float32)
    mid = (0.5 * (z[:, 1:] + z[:, :-1])).astype(np.float32)
    wf = np.zeros((R, S), np.float32)
    wf[:, 1:-1] = w
    for det, uu, ref_key in ((True, None, "det"), (False, u, "rand")):
        zs, zm, order, zstd = sample_pdf_merge(T(z), T(wf), Ns, det=det, u=T(uu) if uu is not None else None, want_order=True)
        zs, zm, order, zstd = N(zs), N(zm), N(order), N(zstd)
        ora = O.sample_pdf(mid, w, Ns, det=det, u=uu)
        # same algorithm, same double-accumulated cdf: the HIP kernel reproduces the oracle to rounding of the lerp
        assert maxabs(zs, ora) < 2e-6
        if np.abs(mid - bins).max() == 0.0:       # mid-points exactly representable: compare with the reference golden
            ulin = np.linspace(0, 1, Ns).astype(np.float32)
            nbad, unexplained = sample_pdf_flip_report(zs, g[f"{ref_key}_{key}"], bins, w, ulin if det else uu)
            assert unexplained == 0 and nbad <= 0.01 * zs.size
        cat = np.concatenate([z, zs], -1)
        assert np.array_equal(zm, np.sort(cat, -1))                      # sortedness
        assert np.array_equal(np.take_along_axis(cat, order.astype(np.int64), -1), zm)   # order is the sort permutation
        assert np.array_equal(np.sort(order, -1), np.tile(np.arange(S + Ns), (R, 1)))
        assert maxabs(zstd, zs.astype(np.float64).std(-1)) < 1e-6


@pytest.mark.parametrize("prec,tol", [("f32", 2e-5), ("f16x3", 2e-5), ("f16", 2e-3), ("bf16", 6e-2)])
@pytest.mark.parametrize("Wd,seed,bias", [(256, 7, True), (256, 8, False), (64, 9, True)])
def test_nerf_mlp_matches_oracle_and_golden(prec, tol, Wd, seed, bias, O):
    """Fused pts + PE + MLP kernel vs NeRF.eval (golden G2 from the reference; oracle for the features)."""
    from evdeblurnerf_amd.nerf import NeRF
    g = load_golden("G2_nerf_mlp")
    tag = {7: "w256", 8: "w256_nobias", 9: "w64"}[seed]
    sd = W.make_nerf_state_dict(seed, W=Wd, rgb_add_bias=bias)
    pts, dirs = g["pts"], g["dirs"]
    n = pts.shape[0]
    # express the golden's free points/dirs as rays with z = 1: o = 0, d = pts, viewdir = dirs
    rb = np.zeros((n, 11), np.float32)
    rb[:, 3:6] = pts
    rb[:, 6], rb[:, 7] = 0.0, 1.0
    rb[:, 8:11] = dirs
    z = np.ones((n, 1), np.float32)
    for ef in ("after_linear", "before_linear"):
        net = NeRF(sd, W=Wd, precision=prec, extract_feature=ef)
        raw, feat = net.mlpforward(T(rb), T(z), want_feature=True)
        raw, feat = N(raw).reshape(n, 4), N(feat).reshape(n, Wd)
        err = maxabs(raw, g[f"raw_{tag}"])
        print(f"[{prec} W={Wd} {tag}] raw L-inf vs reference golden = {err:.3e}")
        assert err < tol
        key = "feat_" if ef == "after_linear" else "featb_"
        assert maxabs(feat[:, :16], g[key + tag]) < tol
    # ragged sizes: 1 sample, and a count that is not a multiple of the workgroup tile
    net = NeRF(sd, W=Wd, precision=prec)
    onet = O.Nerf(sd, W=Wd)
    emb = np.concatenate([O.embed(pts, 10), O.embed(dirs, 4)], -1)
    ref, _, _ = O.nerf_mlp(onet, emb)
    for m in (1, 33, 257, 383):
        raw, _ = net.mlpforward(T(rb[:m]), T(z[:m]))
        assert maxabs(N(raw).reshape(m, 4), ref[:m]) < tol


def _render_cases():
    return [("a", dict(N_samples=64, N_importance=64), 1, 96, 11, 12),
            ("b", dict(N_samples=128, N_importance=0), 2, 80, 13, None),
            ("d", dict(N_samples=64, N_importance=32, perturb=1.0), 4, 48, 11, 12)]


@pytest.mark.parametrize("prec,tol", [("f32", 1e-4), ("f16x3", 1e-4), ("f16", 1e-4), ("bf16", 3e-2)])
def test_render_matches_reference_golden(prec, tol, O):
    """NeRFAll.render end to end vs the goldens produced by the reference (G7). RGB L-inf <= 1e-4 is the
    north-star bound for the float32-grade modes."""
    from types import SimpleNamespace
    from evdeblurnerf_amd.renderer import NeRFAll
    g = load_golden("G7_render_nerf")
    K = W.synthetic_camera()
    for tag, kw, ray_seed, R, sc, sf in _render_cases():
        sd = dict(W.prefixed(W.make_nerf_state_dict(sc), "mlp_coarse"))
        if sf is not None:
            sd.update(W.prefixed(W.make_nerf_state_dict(sf), "mlp_fine"))
        args = SimpleNamespace(mode="nerf", netdepth=8, netwidth=256, multires=10, multires_views=4, use_viewdirs=True,
                               rgb_activate="sigmoid", sigma_activate="relu", N_importance=kw["N_importance"])
        model = NeRFAll(args, sd, precision=prec).eval()
        rays = T(W.synthetic_rays(ray_seed, R))
        extra = {}
        if kw.get("perturb", 0) > 0:
            extra = dict(t_rand=T(g["d_t_rand"]), u=T(g["d_u"]))
        rgb, depth, acc, ex = model.render(400, 400, K, chunk=1 << 20 if tag != "b" else 32, rays=rays, ndc=True, near=0., far=1.,
                                           use_viewdirs=True, retraw=True, raw_noise_std=0., **kw, **extra)
        e_rgb = maxabs(N(rgb), g[f"{tag}_rgb"])
        print(f"[{prec} case {tag}] RGB L-inf vs reference = {e_rgb:.3e}, depth {maxabs(N(depth), g[f'{tag}_depth']):.3e}")
        assert e_rgb < tol
        assert maxabs(N(acc), g[f"{tag}_acc"]) < tol
        assert maxabs(N(depth), g[f"{tag}_depth"]) < 3 * tol
        if kw["N_importance"] > 0:
            assert maxabs(N(ex["rgb0"]), g[f"{tag}_rgb0"]) < tol
            assert maxabs(N(ex["z_vals0"]), g[f"{tag}_z_vals0"]) < 1e-6
            assert maxabs(N(ex["weights0"]), g[f"{tag}_weights0"]) < tol
            frac, worst = z_mismatch(N(ex["z_vals"]), g[f"{tag}_z_vals"], tol={"bf16": 5e-3, "f16": 5e-4}.get(prec, 5e-5))
            assert frac < {"bf16": 0.2, "f16": 0.05}.get(prec, 0.01) and worst < 1.0 / 63 + 1e-3, (frac, worst)
        else:
            assert maxabs(N(ex["z_vals"]), g[f"{tag}_z_vals"]) < 1e-6
            assert maxabs(N(ex["weights"]), g[f"{tag}_weights"]) < tol


@pytest.mark.parametrize("prec,tol", [("f32", 1e-4), ("f16x3", 1e-4), ("f16", 1e-4), ("bf16", 3e-2)])
def test_render_without_viewdirs_matches_reference_golden(prec, tol):
    """use_viewdirs=False (VERDICT r2 missing 3): 8-column ray batch (renderer.py:443-446), output_linear head (nerf.py:158-160; 5 output
    channels with importance sampling, renderer.py:46), the "before_linear" per-sample feature -- against golden G23 from the reference's
    NeRFAll; and what this network is not built for is rejected."""
    from types import SimpleNamespace
    from evdeblurnerf_amd import _lib as L
    from evdeblurnerf_amd.renderer import NeRFAll
    g = load_golden("G23_render_nerf_no_viewdirs")
    K = W.synthetic_camera()
    for tag, Ni, S, R, ndc in (("a", 32, 48, 72, True), ("b", 0, 128, 40, False)):
        och = 5 if Ni > 0 else 4
        sd = dict(W.prefixed(W.make_nerf_state_dict(61, input_ch_views=0, use_viewdirs=False, output_ch=och), "mlp_coarse"))
        if Ni > 0:
            sd.update(W.prefixed(W.make_nerf_state_dict(62, input_ch_views=0, use_viewdirs=False, output_ch=och), "mlp_fine"))
        args = SimpleNamespace(mode="nerf", netdepth=8, netwidth=256, multires=10, multires_views=4, use_viewdirs=False,
                               rgb_activate="sigmoid", sigma_activate="relu", N_importance=Ni, kernel_use_awp=True)
        model = NeRFAll(args, sd, awpnet=object(), precision=prec).eval()
        assert model.extract_feature == "before_linear"
        rays = T(W.synthetic_rays(23 + Ni, R))
        kw = dict(ndc=ndc, near=0. if ndc else 0.5, far=1. if ndc else 3.5, use_viewdirs=False, N_samples=S, N_importance=Ni, retraw=True,
                  perturb=0., raw_noise_std=0.)
        rgb, depth, acc, ex = model.render(400, 400, K, rays=rays, inference=True, **kw)
        e = maxabs(N(rgb), g[f"{tag}_rgb"])
        print(f"[{prec} no viewdirs, case {tag}] RGB L-inf vs reference = {e:.3e}")
        assert e < tol and maxabs(N(acc), g[f"{tag}_acc"]) < tol and maxabs(N(depth), g[f"{tag}_depth"]) < 3 * tol
        assert maxabs(N(ex["rays_d"]), g[f"{tag}_rays_d"]) < 1e-5
        if Ni > 0:
            assert maxabs(N(ex["rgb0"]), g[f"{tag}_rgb0"]) < tol and maxabs(N(ex["z_vals0"]), g[f"{tag}_z_vals0"]) < 1e-6
        else:
            assert maxabs(N(ex["z_vals"]), g[f"{tag}_z_vals"]) < 1e-6 and maxabs(N(ex["weights"]), g[f"{tag}_weights"]) < tol
        if prec in ("f32", "f16x3"):
            feat = model.render(400, 400, K, rays=rays[:2], inference=False, **kw)[3]["depth_feature"]
            assert maxabs(N(feat), g[f"{tag}_depth_feature2"]) < 2e-4
        with pytest.raises(L.EvdError):
            model.render(400, 400, K, rays=rays, use_viewdirs=True, N_samples=S, N_importance=Ni)
    if prec == "f32":
        with pytest.raises(L.EvdError):                  # no compensated-float16 stream, no training path for this network
            NeRFAll(args, sd, awpnet=object(), precision="f16c").eval().render(400, 400, K, rays=rays, **kw)
        with pytest.raises(L.EvdError):
            model.mlp_coarse.mlpforward_train(torch.zeros((4, 8), device=DEV), torch.zeros((4, 16), device=DEV))


@pytest.mark.parametrize("prec,tol", [("f32", 1e-4), ("f16x3", 1e-4), ("f16c", 1e-4), ("f16", 2e-4), ("bf16", 3e-2)])
def test_render_other_multires_matches_reference_golden(prec, tol):
    """--multires / --multires_views other than 10 / 4 (options.py:94-97, embedding.py:101-117): mode='nerf' (6, 2) hierarchical and
    (3, 8) on a 4 x 64 network, mode='c2f' (7, 3) -- against golden G24 from the reference's NeRFAll.  These run in the generic kernels
    (EVD_PREC_F16C falls back to the float32-grade arithmetic there); training them is rejected."""
    from types import SimpleNamespace
    from evdeblurnerf_amd import _lib as L
    from evdeblurnerf_amd.renderer import NeRFAll
    g = load_golden("G24_render_other_multires")
    K = W.synthetic_camera()
    base = dict(mode="nerf", use_viewdirs=True, rgb_activate="sigmoid", sigma_activate="relu")
    # (a)
    Lp, Lv = 6, 2
    sd = dict(W.prefixed(W.make_nerf_state_dict(71, input_ch=W.pe_dim(Lp), input_ch_views=W.pe_dim(Lv)), "mlp_coarse"))
    sd.update(W.prefixed(W.make_nerf_state_dict(72, input_ch=W.pe_dim(Lp), input_ch_views=W.pe_dim(Lv)), "mlp_fine"))
    if prec == "f16c":
        with pytest.raises(L.EvdError):             # the compensated mode of mode='nerf' exists for the (10, 4) 8 x 256 network only
            NeRFAll(SimpleNamespace(netdepth=8, netwidth=256, multires=Lp, multires_views=Lv, N_importance=32, **base), sd, precision=prec).eval().render(
                400, 400, K, rays=T(W.synthetic_rays(41, 56)), ndc=True, near=0., far=1., use_viewdirs=True, N_samples=48, N_importance=32)
    else:
        model = NeRFAll(SimpleNamespace(netdepth=8, netwidth=256, multires=Lp, multires_views=Lv, N_importance=32, **base), sd, precision=prec).eval()
        rgb, depth, acc, ex = model.render(400, 400, K, rays=T(W.synthetic_rays(41, 56)), ndc=True, near=0., far=1., use_viewdirs=True, N_samples=48,
                                           N_importance=32, retraw=True, perturb=0., raw_noise_std=0.)
        e = maxabs(N(rgb), g["a_rgb"])
        print(f"[{prec} multires (6, 2)] RGB L-inf vs reference = {e:.3e}")
        assert e < tol and maxabs(N(ex["rgb0"]), g["a_rgb0"]) < tol and maxabs(N(acc), g["a_acc"]) < tol
        assert maxabs(N(ex["z_vals0"]), g["a_z_vals0"]) < 1e-6
        if prec == "f32":
            with pytest.raises(L.EvdError):
                model.mlp_coarse.mlpforward_train(torch.zeros((4, 11), device=DEV), torch.zeros((4, 16), device=DEV))
        # (b) direction encoding on four k-steps
        Lp, Lv = 3, 8
        sd = W.prefixed(W.make_nerf_state_dict(73, D=4, W=64, input_ch=W.pe_dim(Lp), input_ch_views=W.pe_dim(Lv), skips=()), "mlp_coarse")
        model = NeRFAll(SimpleNamespace(netdepth=4, netwidth=64, multires=Lp, multires_views=Lv, N_importance=0, **base), sd, precision=prec).eval()
        rgb, depth, acc, ex = model.render(400, 400, K, rays=T(W.synthetic_rays(42, 40)), ndc=False, near=0.5, far=3.5, use_viewdirs=True,
                                           N_samples=64, N_importance=0, retraw=True, perturb=0., raw_noise_std=0.)
        e = maxabs(N(rgb), g["b_rgb"])
        print(f"[{prec} multires (3, 8), 4 x 64] RGB L-inf vs reference = {e:.3e}")
        assert e < tol and maxabs(N(ex["weights"]), g["b_weights"]) < tol and maxabs(N(depth), g["b_depth"]) < 3 * tol
    # (c) c2f
    Lp, Lv = 7, 3
    ic, icv = W.pe_dim(Lp), W.pe_dim(Lv)
    a = W.blurfactory_args(32, coarse_voxels=24 ** 3, fine_voxels=48 ** 3)
    a.multires, a.multires_views = Lp, Lv
    gc, gf = W.pdrf_grid_size(*W.BLURFACTORY_AABB, 24 ** 3), W.pdrf_grid_size(*W.BLURFACTORY_AABB, 48 ** 3)
    sd = W.prefixed(W.make_pdrf_state_dict(74, gc, input_ch=32 + ic, input_ch_views=icv, hidden_dim=64, geo_feat_dim=15), "mlp_coarse")
    sd.update(W.prefixed(W.make_pdrf_state_dict(75, gf, input_ch=64 + ic, input_ch_views=icv, hidden_dim=256, geo_feat_dim=128), "mlp_fine"))
    model = NeRFAll(a, sd, precision=prec).eval()
    rgb, depth, acc, ex = model.render(400, 400, K, rays=T(W.synthetic_rays(43, 48)), ndc=True, near=0., far=1., use_viewdirs=True, N_samples=64,
                                       N_importance=32, retraw=True, perturb=0., raw_noise_std=0.)
    e0 = maxabs(N(ex["rgb0"]), g["c_rgb0"])
    same = np.abs(N(ex["z_vals"]) - g["c_z_vals"]).max(-1) < 5e-5           # rays whose importance samples agree (G9's rule)
    e = maxabs(N(rgb)[same], g["c_rgb"][same])
    print(f"[{prec} c2f multires (7, 3)] RGB L-inf vs reference: coarse {e0:.3e}, fine {e:.3e} on {same.mean():.0%} of the rays")
    ctol = tol if prec in ("f32", "f16x3", "f16c") else 10 * tol
    # (single-product modes: the coarse weights move the importance samples of most rays by more than 5e-5; compared where they agree)
    assert e0 < ctol and (same.mean() > 0.8 or prec in ("f16", "bf16")) and (not same.any() or e < ctol)
    if prec == "f32":
        with pytest.raises(L.EvdError):
            model.enable_training(sd).train()
            model(400, 400, K, 1 << 20, rays=T(W.synthetic_rays(43, 48)), ndc=True, near=0., far=1., use_viewdirs=True, N_samples=64, N_importance=32, perturb=1.0)


@pytest.mark.parametrize("prec,tol", [("f32", 1e-4), ("f16x3", 1e-4), ("f16c", 1e-4), ("f16", 5e-4)])
def test_pbe_composite_feature_matches_reference_golden(prec, tol):
    """kernel_type='PBE' (renderer.py:30-34; voxnerf.py:223-239): the coarse level composites its 15 geo features and runs its colour network
    per ray.  Golden G25 from the reference: the level on explicit inputs (composited feature map returned), NeRFAll.render in eval
    mode, NeRFAll.coarse_render (rgb + the feature map the PBE blur kernel consumes).  Training a PBE model is rejected (its blur model
    is out of scope)."""
    from evdeblurnerf_amd import _lib as L
    from evdeblurnerf_amd.renderer import NeRFAll
    g = load_golden("G25_pbe_composite_feature")
    K = W.synthetic_camera()
    a = W.blurfactory_args(32, coarse_voxels=24 ** 3, fine_voxels=48 ** 3)
    a.kernel_type = "PBE"
    gc, gf = W.pdrf_grid_size(*W.BLURFACTORY_AABB, 24 ** 3), W.pdrf_grid_size(*W.BLURFACTORY_AABB, 48 ** 3)
    sd = W.prefixed(W.make_pdrf_state_dict(91, gc, input_ch=95, hidden_dim=64, geo_feat_dim=15), "mlp_coarse")
    sd.update(W.prefixed(W.make_pdrf_state_dict(92, gf, input_ch=127, hidden_dim=256, geo_feat_dim=128), "mlp_fine"))
    model = NeRFAll(a, sd, precision=prec).eval()
    assert model.mlp_coarse.composite_feature and not model.mlp_fine.composite_feature
    ltol = tol if prec != "f16" else 2e-3
    col, dep, acc, wts, fm = model.mlp_coarse(T(g["l_pts"]), T(g["l_vd"]), T(g["l_fts"]), T(g["l_z"]), T(g["l_rd"]))
    assert fm.shape == (24, 15)
    e = {k: maxabs(N(v), g["l_" + k]) for k, v in (("color", col), ("depth", dep), ("acc", acc), ("weights", wts), ("feature", fm))}
    print(f"[{prec} PBE level] " + ", ".join(f"{k} {v:.2e}" for k, v in e.items()))
    assert all(v < ltol for v in e.values()), e
    rays = T(W.synthetic_rays(51, 56))
    crgb, cfeat = model.coarse_render(400, 400, K, rays=rays, ndc=True, near=0., far=1., use_viewdirs=True, N_samples=64, N_importance=32,
                                      perturb=0., raw_noise_std=0.)
    assert cfeat.shape == (56, 15)
    assert maxabs(N(crgb), g["coarse_rgb"]) < ltol and maxabs(N(cfeat), g["coarse_feat"]) < ltol
    rgb, depth, acc, ex = model.render(400, 400, K, rays=rays, ndc=True, near=0., far=1., use_viewdirs=True, N_samples=64, N_importance=32,
                                       retraw=True, perturb=0., raw_noise_std=0.)
    e0 = maxabs(N(ex["rgb0"]), g["rgb0"])
    same = np.abs(N(ex["z_vals"]) - g["z_vals"]).max(-1) < 5e-5
    efine = maxabs(N(rgb)[same], g["rgb"][same]) if same.any() else 0.0
    print(f"[{prec} PBE render] rgb0 {e0:.2e}, rgb {efine:.2e} on {same.mean():.0%} of the rays")
    assert e0 < ltol and efine < ltol and (same.mean() > 0.8 or prec == "f16")
    assert maxabs(N(ex["z_vals0"]), g["z_vals0"]) < 1e-6
    # mode='nerf' with PBE: coarse_render composites the coarse network's per-sample feature (nerf.py:167-169)
    from types import SimpleNamespace
    an = SimpleNamespace(mode="nerf", netdepth=8, netwidth=256, multires=10, multires_views=4, use_viewdirs=True, rgb_activate="sigmoid",
                         sigma_activate="relu", N_importance=0, kernel_type="PBE")
    if prec != "f16c":
        mn = NeRFAll(an, W.prefixed(W.make_nerf_state_dict(93), "mlp_coarse"), precision=prec).eval()
        nrgb, nfeat = mn.coarse_render(400, 400, K, rays=rays[:24], ndc=True, near=0., far=1., use_viewdirs=True, N_samples=48, perturb=0., raw_noise_std=0.)
        assert nfeat.shape == (24, 256)
        assert maxabs(N(nrgb), g["nerf_coarse_rgb"]) < ltol and maxabs(N(nfeat), g["nerf_coarse_feat"]) < (ltol if prec != "f16" else 2e-2)
    # the composed map through raw2outputs on a 16-channel raw (voxnerf.py:223-229)
    raw16 = torch.cat([T(g["l_weights"])[..., None] * 0 + 1.0, torch.rand((24, 40, 15), device=DEV)], -1)
    fmap, dens, acc2, wts2, dep2 = model.mlp_coarse.raw2outputs(raw16, T(g["l_z"]), T(g["l_rd"]))
    assert fmap.shape == (24, 15) and torch.allclose((wts2[..., None] * torch.relu(raw16[..., 1:])).sum(1), fmap, atol=1e-5)
    if prec == "f32":
        with pytest.raises((L.EvdError, NotImplementedError)):
            model.enable_training(sd).train()
            model(400, 400, K, 1 << 20, rays=rays, ndc=True, near=0., far=1., use_viewdirs=True, N_samples=64, N_importance=32, perturb=1.0)


def test_render_config1_white_bkgd_lindisp_no_ndc(O):
    """BASELINE config 1 shape (single pass, 64 samples) with the non-default switches, vs golden G7c."""
    from types import SimpleNamespace
    from evdeblurnerf_amd.renderer import NeRFAll
    g = load_golden("G7_render_nerf")
    sd = W.prefixed(W.make_nerf_state_dict(13), "mlp_coarse")
    args = SimpleNamespace(mode="nerf", netdepth=8, netwidth=256, multires=10, multires_views=4, use_viewdirs=True,
                           rgb_activate="sigmoid", sigma_activate="relu", N_importance=0)
    model = NeRFAll(args, sd, precision="f16x3").eval()
    rgb, depth, acc, ex = model.render(400, 400, W.synthetic_camera(), rays=T(W.synthetic_rays(3, 64)), ndc=False, near=0.5, far=3.5,
                                       use_viewdirs=True, N_samples=64, N_importance=0, retraw=False, perturb=0., raw_noise_std=0.,
                                       white_bkgd=True, lindisp=True)
    assert maxabs(N(rgb), g["c_rgb"]) < 1e-4
    assert maxabs(N(depth), g["c_depth"]) < 3e-4
    assert set(ex.keys()) == set()
    # empty batch: the reference returns empty tensors (renderer.py:450 "max(1, .)")
    rgb, depth, acc, ex = model.render(400, 400, W.synthetic_camera(), rays=torch.empty((0, 3, 2), device=DEV), ndc=True, near=0., far=1.,
                                       use_viewdirs=True, N_samples=64, N_importance=0, retraw=True)
    assert rgb.shape == (0, 3) and ex["weights"].shape == (0, 64)


def test_full_size_metric_config_properties():
    """BASELINE.json metric shape: 4096 rays x 128 samples through the 8x256 MLP. Size-independent properties:
    acc == 1 (last alpha forced to 1), weights >= 0 and sum to acc, rgb in [0,1], depth within [near,far],
    ray-permutation equivariance, chunk invariance, and agreement of the four arithmetic modes."""
    from types import SimpleNamespace
    from evdeblurnerf_amd.renderer import NeRFAll
    sd = W.prefixed(W.make_nerf_state_dict(21), "mlp_coarse")
    args = SimpleNamespace(mode="nerf", netdepth=8, netwidth=256, multires=10, multires_views=4, use_viewdirs=True,
                           rgb_activate="sigmoid", sigma_activate="relu", N_importance=0)
    rays = T(W.synthetic_rays(7, 4096))
    K = W.synthetic_camera()
    kw = dict(ndc=True, near=0., far=1., use_viewdirs=True, N_samples=128, N_importance=0, retraw=True)
    out = {}
    for prec in ("f32", "f16x3", "f16", "bf16"):
        model = NeRFAll(args, sd, precision=prec).eval()
        rgb, depth, acc, ex = model.render(400, 400, K, rays=rays, **kw)
        out[prec] = N(rgb)
        assert np.allclose(N(acc), 1.0, atol=2e-5)
        w = N(ex["weights"])
        assert (w >= -1e-7).all() and np.allclose(w.sum(-1), N(acc), atol=2e-5)
        assert out[prec].min() >= 0.0 and out[prec].max() <= 1.0
        assert N(depth).min() >= -1e-6 and N(depth).max() <= 1.0 + 1e-5
        if prec == "f16x3":
            perm = torch.randperm(4096, device=DEV)
            rgb_p = model.render(400, 400, K, rays=rays[perm], **kw)[0]
            assert torch.equal(rgb_p, rgb[perm])                      # rays are independent
            rgb_c = model.render(400, 400, K, chunk=1000, rays=rays, **kw)[0]
            assert torch.equal(rgb_c, rgb)                            # chunking does not change results (renderer.py:406)
    assert maxabs(out["f16x3"], out["f32"]) < 1e-4
    print(f"f16 vs f32 RGB L-inf at full size: {maxabs(out['f16'], out['f32']):.3e}")
    assert maxabs(out["f16"], out["f32"]) < 1e-4        # single-product float16 (software-pipelined kernel) on these weights
    print(f"bf16 vs f32 RGB L-inf at full size: {maxabs(out['bf16'], out['f32']):.3e}")
    assert maxabs(out["bf16"], out["f32"]) < 3e-2


def test_weighted_sum_and_crf_and_losses(O):
    from evdeblurnerf_amd.losses import (weighted_sum, rbk_weighted_sum, blur_loss_partials, blur_loss_from_partials,
                                          event_loss_partials, event_loss_from_partials, egm_loss, img2mse)
    from evdeblurnerf_amd.tonemapping import TonemappingTransform, CRF
    g = load_golden("G10_rbk_weighted_sum")
    ccw = T(g["ccw"])
    o_rgb, o_depth, o_acc, o_ex = rbk_weighted_sum(T(g["rgb"]), T(g["depth"]), T(g["acc"]),
                                                   {k: T(g["ex_" + k]) for k in ("rgb0", "z_std", "weights", "depth_feature")}, ccw)
    assert maxabs(N(o_rgb), g["o_rgb"]) < 1e-6 and maxabs(N(o_depth), g["o_depth"]) < 1e-6 and maxabs(N(o_acc), g["o_acc"]) < 1e-6
    for k, v in o_ex.items():
        assert maxabs(N(v), g["o_" + k]) < 1e-6, k

    g = load_golden("G11_crf")
    x, f2, f32 = T(g["x"]), T(g["f2"]), T(g["f32"])
    sd = dict(W.prefixed(W.make_crf_state_dict(41, 2), "tonemapping_event"))
    tm = TonemappingTransform("gamma", "learn", state_dict=sd, extra_features_event=2)
    assert maxabs(N(tm(x, mode="encode_rgb")), g["rgb_gamma"]) < 2e-6
    assert maxabs(N(tm(x, mode="encode_luma", ev_extra_feat=f2)), g["luma_learn_f2"]) < 2e-6
    assert maxabs(N(tm(x, mode="encode_luma")), g["luma_learn_nofeat"]) < 2e-6
    assert maxabs(N(tm(x, mode="encode_luma", skip_learn_crf=True, ev_extra_feat=f2)), g["luma_learn_skip"]) < 2e-6
    assert maxabs(N(tm(x, mode="encode_luma", tonemap_only=True, ev_extra_feat=f32)), g["tone_learn_f32"]) < 2e-6
    assert maxabs(N(tm(x, mode="encode_luma", keep_rgb=True, ev_extra_feat=f2)), g["luma_learn_keep"]) < 2e-6
    sd0 = dict(W.prefixed(W.make_crf_state_dict(43, 0), "tonemapping_event"))
    tm0 = TonemappingTransform("none", "learn", state_dict=sd0)
    assert maxabs(N(tm0(x, mode="encode_rgb")), g["rgb_none"]) == 0.0
    assert maxabs(N(tm0(x, mode="encode_luma")), g["luma_learn0"]) < 2e-6
    for std in ("rec601", "rec709", "avg"):
        tmg = TonemappingTransform("gamma", "gamma", luma_standard=std)
        key = "luma_gamma" if std == "rec601" else f"luma_gamma_{std}"
        assert maxabs(N(tmg(x, mode="encode_luma")), g[key]) < 2e-6

    g = load_golden("G12_egm_loss")
    rel = lambda a, b: abs(float(a) - float(b)) / max(abs(float(b)), 1e-12)
    assert rel(egm_loss(T(g["ls"]), T(g["le"]), T(g["bii"])), g["loss_plain"]) < 1e-5
    assert rel(egm_loss(T(g["ls3"]), T(g["le3"]), T(g["bii"]), color_mask=T(g["cmask"])), g["loss_mask"]) < 1e-5
    assert rel(egm_loss(T(g["ls3"]), T(g["le3"]), T(g["bii"]), color_mask=T(g["cmask"]), color_weight=[0.4, 0.2, 0.4]), g["loss_mask_w"]) < 1e-5
    assert rel(img2mse(T(g["ls3"]), T(g["le3"])), O.mse(g["ls3"], g["le3"])) < 1e-5

    g = load_golden("G14_loss_assembly")
    for cfg in ("blender", "cdavis"):
        flw, w_pts0, w_egm = [float(v) for v in g[f"{cfg}_scalars"]]
        crf_rgb = CRF("gamma" if cfg == "blender" else "none")
        crf_ev = CRF("learn", state_dict=W.make_crf_state_dict(51, 2), extra_features=2)
        ccw = g[f"{cfg}_ccw"]
        p, cols = blur_loss_partials(crf_rgb, T(g[f"{cfg}_rgb_p"]), T(ccw[0]), T(g[f"{cfg}_target"]), rgb0_p=T(g[f"{cfg}_rgb0_p"]),
                                     w2=T(ccw[1]), target_pts0=T(g[f"{cfg}_target_pts0"]), want_colours=True)
        loss, terms = blur_loss_from_partials(p, fine_loss_weight=flw, w_pts0=w_pts0)
        assert abs(float(loss) - float(g[f"{cfg}_img_loss"])) < 2e-6
        assert abs(float(terms["pts0"]) - float(g[f"{cfg}_pts0"])) < 2e-6
        assert maxabs(N(cols["rgb"]), O.weighted_sum(g[f"{cfg}_rgb_p"], ccw[0])) < 1e-6
        thr = 0.2 if cfg == "blender" else 0.25
        kw = dict(add_bii="pos-neg") if cfg == "blender" else dict(add_bii="color-pos-neg", tonemap_only=True,
                                                                   color_mask=T(g[f"{cfg}_cmask"]), color_weight=[0.4, 0.2, 0.4])
        pe = event_loss_partials(crf_ev, T(g[f"{cfg}_es"]), T(g[f"{cfg}_ee"]), T(g[f"{cfg}_cn"]), T(g[f"{cfg}_cp"]), thr, thr,
                                 start0=T(g[f"{cfg}_es0"]), end0=T(g[f"{cfg}_ee0"]), **kw)
        egm = event_loss_from_partials(pe)
        assert rel(egm, g[f"{cfg}_egm"]) < 2e-5
        total = float(loss) + float(egm) * w_egm
        assert abs(total - float(g[f"{cfg}_total"])) < 1e-5 * max(1.0, float(g[f"{cfg}_total"]))
    # ragged event counts (the kernel packs 16 events per block): partial sums are additive over any split of the batch
    rs = np.random.RandomState(6)
    ev = lambda n: T(rs.rand(n, 3).astype(np.float32) * 0.9 + 0.05)
    es, ee, es0, ee0 = ev(37), ev(37), ev(37), ev(37)
    cn, cp = T(-rs.randint(0, 4, 37).astype(np.float32)), T(rs.randint(0, 4, 37).astype(np.float32))
    full = N(event_loss_partials(crf_ev, es, ee, cn, cp, 0.2, 0.2, start0=es0, end0=ee0))
    parts = np.zeros_like(full)
    for lo, hi in ((0, 1), (1, 18), (18, 37)):
        parts += N(event_loss_partials(crf_ev, es[lo:hi], ee[lo:hi], cn[lo:hi], cp[lo:hi], 0.2, 0.2, start0=es0[lo:hi], end0=ee0[lo:hi]))
    assert np.allclose(full, parts, rtol=1e-5, atol=1e-6) and full[2] == 37.0
    # linearity of the sub-exposure reduction (size-independent property, config-3 shape: 1024 px x P=10)
    rs = np.random.RandomState(5)
    xa, xb = T(rs.rand(10240, 3).astype(np.float32)), T(rs.rand(10240, 3).astype(np.float32))
    cc = T(rs.dirichlet(np.ones(10), 1024).astype(np.float32))
    assert maxabs(N(weighted_sum(xa + xb, cc)), N(weighted_sum(xa, cc) + weighted_sum(xb, cc))) < 1e-6


def test_edi_matches_golden():
    from evdeblurnerf_amd.edi import brightness_increment_image, deblur_double_integral
    g = load_golden("G13_edi")
    for s in range(8):
        img = brightness_increment_image(T(g[f"x{s}"]), T(g[f"y{s}"]), T(g[f"p{s}"]), 16, 16, 0.2, 0.25, True)
        assert maxabs(N(img), g["bii"][s]) < 2e-6
        img = brightness_increment_image(T(g[f"x{s}"]), T(g[f"y{s}"]), T(g[f"p{s}"]), 16, 16, 0.2, 0.25, False)
        assert maxabs(N(img), g[f"bii_ni{s}"]) < 2e-6
    assert maxabs(N(deblur_double_integral(T(g["blurry"]), T(g["bii"]))), g["sharp"]) < 2e-6
    assert maxabs(N(deblur_double_integral(T(g["blurry3"]), T(g["bii3"]))), g["sharp3"]) < 2e-6


# ----------------------------------------------------------------------------------------------- PDRF (mode='c2f')
AABB = ([-1.5, -1.5, -1.0], [1.5, 1.5, 1.0])


def _c2f_args(N_importance=64):
    from types import SimpleNamespace
    return SimpleNamespace(mode="c2f", multires=10, multires_views=4, use_viewdirs=True, N_importance=N_importance,
                           kernel_type="RBK", kernel_use_awp=False, rgb_activate="sigmoid", sigma_activate="relu",
                           bounding_box=AABB, coarse_num_layers=2, coarse_num_layers_color=3, coarse_hidden_dim=64,
                           coarse_hidden_dim_color=64, coarse_app_dim=32, coarse_app_n_comp=[64, 16, 16], coarse_n_voxels=24 ** 3,
                           kernel_feat_cnl=15, fine_num_layers=2, fine_num_layers_color=3, fine_hidden_dim=256,
                           fine_hidden_dim_color=256, fine_geo_feat_dim=128, fine_app_dim=32, fine_app_n_comp=[64, 16, 16],
                           fine_n_voxels=48 ** 3)


def _c2f_sd(seed_c, seed_f):
    gc = W.pdrf_grid_size(AABB[0], AABB[1], 24 ** 3)
    gf = W.pdrf_grid_size(AABB[0], AABB[1], 48 ** 3)
    sd = dict(W.prefixed(W.make_pdrf_state_dict(seed_c, gc, input_ch=95, hidden_dim=64, geo_feat_dim=15), "mlp_coarse"))
    sd.update(W.prefixed(W.make_pdrf_state_dict(seed_f, gf, input_ch=127, hidden_dim=256, geo_feat_dim=128), "mlp_fine"))
    return sd, gc, gf


def test_voxel_sample_matches_golden(O):
    from evdeblurnerf_amd.renderer import NeRFAll
    g = load_golden("G8_appfeature")
    sd, gc, gf = _c2f_sd(21, 22)
    model = NeRFAll(_c2f_args(), sd).eval()
    assert model.mlp_coarse.gridSize == list(g["grid_coarse"]) and model.mlp_fine.gridSize == list(g["grid_fine"])
    pts = T(g["pts"])
    # bilinear taps + basis matmul in the ATen-CPU operation order: float32 rounding only
    assert maxabs(N(model.mlp_coarse.sample(pts)), g["ft_coarse"]) < 2e-6
    assert maxabs(N(model.mlp_fine.sample(pts)), g["ft_fine"]) < 2e-6
    # large batch incl. points far outside the box (zero padding) against the oracle
    rs = np.random.RandomState(3)
    big = (rs.uniform(-2.2, 2.2, size=(20000, 3)) * np.array([1.0, 1.0, 0.7])).astype(np.float32)
    aabb6 = AABB[0] + AABB[1]
    vc = O.Voxel(sd, "mlp_coarse.", gc, aabb6, input_ch=95)
    assert maxabs(N(model.mlp_coarse.sample(T(big))), O.appfeature(vc, big)) < 2e-6
    # TV regulariser (voxnerf.py:126-130) vs the oracle's TVLoss restatement
    tv_ref = sum(O.tv_loss(sd[f"mlp_coarse.app_plane.{i}"]) * 1e-2 + O.tv_loss(sd[f"mlp_coarse.app_line.{i}"]) * 1e-3 for i in range(3))
    assert abs(float(model.mlp_coarse.TV_loss_app()) - tv_ref) < 1e-5 * tv_ref


@pytest.mark.parametrize("prec,tol", [("f32", 1e-4), ("f16x3", 1e-4), ("f16", 2e-4), ("bf16", 3e-2)])
def test_c2f_render_matches_reference_golden(prec, tol, O):
    """NeRFAll.render mode='c2f' (PDRF coarse + fine level) vs goldens produced by the reference (G9)."""
    from evdeblurnerf_amd.renderer import NeRFAll
    g = load_golden("G9_render_c2f")
    sd, gc, gf = _c2f_sd(31, 32)
    K = W.synthetic_camera()
    rays = T(W.synthetic_rays(9, 64))
    kw = dict(ndc=True, near=0., far=1., use_viewdirs=True, N_samples=64, retraw=True, perturb=0., raw_noise_std=0.)
    model = NeRFAll(_c2f_args(), sd, precision=prec).eval()
    rgb, depth, acc, ex = model.render(400, 400, K, rays=rays, N_importance=64, **kw)
    e = maxabs(N(rgb), g["rgb"])
    print(f"[c2f {prec}] RGB L-inf vs reference = {e:.3e}; coarse rgb0 {maxabs(N(ex['rgb0']), g['rgb0']):.3e}")
    assert e < tol
    assert maxabs(N(ex["rgb0"]), g["rgb0"]) < tol
    # depth follows the resampled z positions, which move when a coarse weight changes in its 4th digit: float16 gets 2e-3
    assert maxabs(N(acc), g["acc"]) < tol and maxabs(N(depth), g["depth"]) < (2e-3 if prec == "f16" else 3 * tol)
    assert maxabs(N(ex["z_vals0"]), g["z_vals0"]) < 1e-6
    assert maxabs(N(ex["weights0"]), g["weights0"]) < tol
    frac, worst = z_mismatch(N(ex["z_vals"]), g["z_vals"], tol={"bf16": 5e-3, "f16": 5e-4}.get(prec, 5e-5))
    assert frac < {"bf16": 0.3, "f16": 0.05}.get(prec, 0.01) and worst < 1.0 / 63 + 1e-3, (frac, worst)
    model0 = NeRFAll(_c2f_args(0), sd, precision=prec).eval()
    rgb, depth, acc, ex = model0.render(400, 400, K, rays=rays, N_importance=0, **kw)
    assert maxabs(N(rgb), g["c_rgb"]) < tol and maxabs(N(ex["weights"]), g["c_weights"]) < tol
    if prec == "f16x3":
        # per-sample fine-level features (AWP input) and the NDC ray directions returned with them
        from types import SimpleNamespace
        model.use_awp = True
        rgb, depth, acc, ex = model.render(400, 400, K, rays=rays[:16], N_importance=64, **kw)
        assert ex["depth_feature"].shape == (16, 128, 128)
        tight = np.abs(N(ex["z_vals"]) - g["z_vals"][:16]).max(-1) < 2e-6
        assert tight.sum() >= 4
        assert maxabs(N(ex["depth_feature"])[:, :, :8][tight], g["f_depth_feature"][tight]) < 2e-4
        assert maxabs(N(ex["rays_d"]), g["f_rays_d"]) < 2e-6
        # VoxelNeRFBase.forward entry on its own, against the oracle
        aabb6 = AABB[0] + AABB[1]
        vf = O.Voxel(sd, "mlp_fine.", gf, aabb6, input_ch=127, hidden_dim=256, geo_feat_dim=128, rgb_act="none")
        rs = np.random.RandomState(4)
        R, S = 24, 40
        pts = rs.uniform(-1, 1, size=(R, S, 3)).astype(np.float32)
        vd = rs.standard_normal((R, 3)).astype(np.float32)
        vd /= np.linalg.norm(vd, axis=-1, keepdims=True)
        fts = rs.standard_normal((R, S, 64)).astype(np.float32) * 0.3
        z = np.sort(rs.uniform(0, 1, size=(R, S)).astype(np.float32), -1)
        rd = rs.standard_normal((R, 3)).astype(np.float32)
        col, dep, ac, wt, ft = model.mlp_fine(T(pts), T(vd), T(fts), T(z), T(rd))
        import ctypes as C
        oc, od, oa, ow = (np.empty((R, 3), np.float32), np.empty((R,), np.float32), np.empty((R,), np.float32), np.empty((R, S), np.float32))
        of = np.empty((R, S, 128), np.float32)
        fpp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        O.lib().evo_voxel_forward(C.byref(vf.s), fpp(pts), fpp(vd), fpp(fts), 64, fpp(z), fpp(rd), C.c_long(R), S, 10, 4, 0,
                                  fpp(oc), fpp(od), fpp(oa), fpp(ow), fpp(of))
        assert maxabs(N(col), oc) < 2e-5 and maxabs(N(wt), ow) < 2e-5 and maxabs(N(ft), of) < 5e-5


def test_awp_feature_integration(O):
    """The AWP consumer's compositing scan (awp.py:49-77) vs the reference golden G15 and, at the blurfactory shape
    (10 240 rays x 128 samples x 64 channels), vs the oracle on a slice + ray-permutation equivariance."""
    from evdeblurnerf_amd.awp import feature_integration
    g = load_golden("G15_awp_feature_integration")
    for tag in ("a", "b"):
        out = feature_integration(T(g[f"{tag}_feat"]), T(g[f"{tag}_z"]), T(g[f"{tag}_rays_d"]))
        assert maxabs(N(out), g[f"{tag}_out"]) < 2e-5 * max(1.0, np.abs(g[f"{tag}_out"]).max())
    rs = np.random.RandomState(8)
    R, P, S, Cc = 1024, 10, 128, 64
    feat = T(np.abs(rs.standard_normal((R, P, S, Cc))).astype(np.float32))
    z = T(np.sort(rs.uniform(0, 1, (R * P, S)).astype(np.float32), -1))
    rd = T(rs.standard_normal((R * P, 3)).astype(np.float32))
    out = N(feature_integration(feat, z, rd))
    ref = O.awp_feature_integration(N(feat).reshape(-1, S, Cc)[:640], N(z)[:640], N(rd)[:640])
    assert maxabs(out.reshape(-1, Cc)[:640], ref) < 1e-5
    perm = torch.randperm(R * P, device=DEV)      # rays are independent: permutation equivariance at the full shape
    out_p = feature_integration(feat.reshape(R * P, 1, S, Cc)[perm], z[perm], rd[perm])
    assert torch.equal(out_p.reshape(-1, Cc), torch.as_tensor(out, device=DEV).reshape(-1, Cc)[perm])
    wide = T(np.abs(rs.standard_normal((5, 2, 40, 128))).astype(np.float32))     # 128 channels: two per lane
    zw, dw = T(np.sort(rs.uniform(0, 1, (10, 40)).astype(np.float32), -1)), T(rs.standard_normal((10, 3)).astype(np.float32))
    assert maxabs(N(feature_integration(wide, zw, dw)).reshape(-1, 128), O.awp_feature_integration(N(wide).reshape(-1, 40, 128), N(zw), N(dw))) < 1e-5


def test_rbk_warp(O):
    """Sub-exposure ray warp (blurmodel.py:51-82) vs the reference golden G16 and, at the blur-batch shape (1024 px x 9
    motions), vs the oracle; rigid transforms are orthonormal (size-independent property)."""
    from evdeblurnerf_amd.rays import rbk_warp
    g = load_golden("G16_rbk_warp")
    for tag, M, uo in (("a", 9, True), ("b", 4, False), ("c", 9, True)):
        new_rays, tf = rbk_warp(T(g[f"{tag}_rays"]), T(g[f"{tag}_r"]), T(g[f"{tag}_v"]), M, uo, return_transform=True)
        assert maxabs(N(new_rays), g[f"{tag}_new_rays"]) < 5e-6, tag
        assert maxabs(N(tf), g[f"{tag}_transform"]) < 5e-6, tag
    rs = np.random.RandomState(9)
    rays = W.synthetic_rays(77, 1024)
    r, v = (rs.standard_normal((1024, 27)) * 0.05).astype(np.float32), (rs.standard_normal((1024, 27)) * 0.05).astype(np.float32)
    new_rays, tf = rbk_warp(T(rays), T(r), T(v), 9, True, return_transform=True)
    assert maxabs(N(new_rays), O.rbk_warp(rays, r, v, 9, True)) < 5e-6
    Rm = N(tf)[..., :3, :3]
    assert maxabs(Rm @ np.swapaxes(Rm, -1, -2), np.broadcast_to(np.eye(3, dtype=np.float32), Rm.shape)) < 1e-5
    assert rbk_warp(torch.empty((0, 3, 2), device=DEV), torch.empty((0, 27), device=DEV), torch.empty((0, 27), device=DEV), 9).shape == (0, 10, 3, 2)


def test_full_frame_eval_path_psnr_parity(O):
    """BASELINE config 5 in small: NeRFAll in eval mode (forward -> render_path -> get_rays -> render, renderer.py:394-397,
    594-626) renders whole frames with hierarchical 24 + 40 samples and render_kwargs_test; against the oracle's render of
    the same rays: RGB L-inf <= 1e-4 (f16x3) and PSNR between the two frames >= 80 dB; f16 mode PSNR >= 70 dB."""
    from types import SimpleNamespace
    from evdeblurnerf_amd.renderer import NeRFAll
    Hh, Ww = 20, 28
    K = W.synthetic_camera(Hh, Ww, 30.0)
    sd = dict(W.prefixed(W.make_nerf_state_dict(11), "mlp_coarse"))
    sd.update(W.prefixed(W.make_nerf_state_dict(12), "mlp_fine"))
    args = SimpleNamespace(mode="nerf", netdepth=8, netwidth=256, multires=10, multires_views=4, use_viewdirs=True,
                           rgb_activate="sigmoid", sigma_activate="relu", N_importance=40)
    poses = [W.synthetic_pose(60 + i) for i in range(2)]
    kw = dict(ndc=True, near=0., far=1., use_viewdirs=True, N_samples=24, N_importance=40, perturb=0., raw_noise_std=0.)
    ref = []
    for c2w in poses:
        o, d = O.get_rays(Hh, Ww, K, c2w)
        rays = np.stack([o, d], -1).reshape(-1, 3, 2)
        cfg = O.make_cfg(H=Hh, W=Ww, focal=float(K[0][0]), N_samples=24, N_importance=40)
        ref.append(O.render_nerf(O.Nerf(sd, "mlp_coarse."), O.Nerf(sd, "mlp_fine."), cfg, rays)["rgb"].reshape(Hh, Ww, 3))
    ref = np.stack(ref)
    psnr = lambda a, b: -10.0 * np.log10(max(float(((a - b) ** 2).mean()), 1e-20))
    for prec, linf, db in (("f16x3", 1e-4, 80.0), ("f16", 1e-3, 70.0)):
        model = NeRFAll(args, sd, precision=prec).eval()
        rgbs, depths = model(Hh, Ww, K, poses=[torch.as_tensor(p) for p in poses], render_kwargs=kw)
        assert rgbs.shape == (2, Hh, Ww, 3) and depths.shape == (2, Hh, Ww)
        got = N(rgbs)
        print(f"[full frame {prec}] L-inf {maxabs(got, ref):.2e}  PSNR vs oracle frame {psnr(got, ref):.1f} dB")
        assert maxabs(got, ref) < linf and psnr(got, ref) > db
        sharded = model.render_path(Hh, Ww, K, 1 << 20, [torch.as_tensor(poses[0])], kw, shard_rows=True)[0]   # no process group: identity
        assert torch.equal(sharded[0], rgbs[0])
        if prec == "f16x3":
            # render(c2w=...) without rays generates the image's rays on the device; with rays, c2w is not read (renderer.py:423)
            a = model.render(Hh, Ww, K, c2w=poses[0], **kw)[0]
            assert a.shape == (Hh, Ww, 3) and torch.equal(a.cpu(), rgbs[0].cpu())
            o, d = O.get_rays(Hh, Ww, K, poses[1])
            r1 = T(np.stack([o, d], -1))
            assert torch.equal(model.render(Hh, Ww, K, rays=r1, c2w=poses[0], **kw)[0].cpu(), rgbs[1].cpu())
            # c2w_staticcam (renderer.py:427-430): camera 0's origins / directions, view directions of camera 1's rays
            sc = model.render(Hh, Ww, K, rays=r1, c2w_staticcam=poses[0], **kw)[0]
            o0, d0 = O.get_rays(Hh, Ww, K, poses[0])
            rb = NeRFAll.ray_batch_train(Hh, Ww, K, T(np.stack([o0, d0], -1).reshape(-1, 3, 2)))
            vd = r1.reshape(-1, 3, 2)[..., 1]
            rb[:, 8:11] = vd / vd.norm(dim=-1, keepdim=True)
            want = model.render_rays(rb, 24, N_importance=40)["rgb_map"].reshape(Hh, Ww, 3)
            assert maxabs(N(sc), N(want)) < 2e-5
            from evdeblurnerf_amd import _lib as L_
            with pytest.raises(L_.EvdError):                     # a model built with the view branch takes the 11-column batch only
                model.render(Hh, Ww, K, rays=r1, use_viewdirs=False, N_samples=24)


def test_compute_successor_bit_exact(O):
    """Event successor graph (utils/events.py:72-120): bit-exact vs the reference golden G17, vs the oracle on 2 M events of a
    346x260 sensor, plus the chain property (following successors visits a pixel's events in increasing order)."""
    from evdeblurnerf_amd.events import compute_successor
    g = load_golden("G17_compute_successor")
    for tag in ("a", "b", "c"):
        hw = g[f"{tag}_latest"].shape[0]
        succ, nsucc, latest, first = compute_successor(T(g[f"{tag}_ids"]), hw)
        assert np.array_equal(N(succ), g[f"{tag}_succ"]) and np.array_equal(N(nsucc), g[f"{tag}_nsucc"]), tag
        assert np.array_equal(N(latest), g[f"{tag}_latest"]) and np.array_equal(N(first), g[f"{tag}_first"]), tag
    rs = np.random.RandomState(10)
    hw, n = 346 * 260, 2_000_000
    ids = rs.randint(0, hw, size=n).astype(np.int32)
    succ, nsucc, latest, first = [N(v) for v in compute_successor(T(ids), hw)]
    rs_, rn, rl, rf = O.compute_successor(ids, hw)
    assert np.array_equal(succ, rs_) and np.array_equal(nsucc, rn) and np.array_equal(latest, rl) and np.array_equal(first, rf)
    assert (succ >= np.arange(n)).all() and (ids[succ] == ids).all()
    empty = compute_successor(torch.empty((0,), dtype=torch.int32, device=DEV), 7)
    assert empty[0].shape == (0,) and (N(empty[2]) == -1).all()


def test_sample_events_matches_golden_and_oracle(O):
    """Event batch assembly (EventsDataset.sample_events, data/loader_events.py:259-304) in one launch on resident tables: golden G26
    (the reference's gather_successor / get_rays_pix composed as the method composes them) -- polarity sums, ids, colour maps
    bit-exact, rays within 1e-6; the oracle on 2 M events / a 65 536-event batch bit for bit (same unfused float32 arithmetic);
    properties: the end event lies on the start event's coordinate, hop count = chain distance; an empty batch; the mismatch flag."""
    from evdeblurnerf_amd.events import EventSampler, compute_successor
    from test_oracle_golden import _check_sample_events
    g = load_golden("G26_sample_events")

    def run(tag, hops, K):
        smp = EventSampler(g[f"{tag}_events"], g[f"{tag}_coords"], g[f"{tag}_poses"], K, id_to_color_map=g[f"{tag}_cmap"] if f"{tag}_cmap" in g else None,
                           integer_coords=bool(g[f"{tag}_halfpix"]))
        out = smp.sample_events(T(g[f"{tag}_ids"]), hops=T(hops) if hops is not None else None, check=True)
        return {k: (N(v) if v is not None else None) for k, v in out.items()}
    _check_sample_events(run, g, exact_rays=False)
    # full size: 2 M events on a 346 x 260 sensor, batch of 65 536
    rs = np.random.RandomState(12)
    hw, n, nq = 346 * 260, 2_000_000, 65536
    ids = rs.randint(0, hw, size=n).astype(np.int32)
    succ, nsucc, _, _ = [N(v) for v in compute_successor(T(ids), hw)]
    events = np.stack([ids.astype(np.float64), np.sort(rs.uniform(0, 1e7, n)), rs.choice([-1.0, 1.0], n), succ.astype(np.float64)], -1)
    coords = np.stack([np.arange(hw) % 346, np.arange(hw) // 346], -1).astype(np.float32)
    poses = rs.standard_normal((n, 3, 4)).astype(np.float32)
    K = W.synthetic_camera()
    smp = EventSampler(events, coords, poses, K)
    q = np.where(nsucc > 0)[0][rs.randint(0, int((nsucc > 0).sum()), nq)]
    hops = np.minimum(rs.randint(0, 9, nq), nsucc[q] - 1)
    hops[:50] = nsucc[q[:50]] + 3                                      # beyond the chain: the last event is its own successor, sums keep adding it
    for hp in (None, hops):
        got = {k: N(v) for k, v in smp.sample_events(T(q), hops=T(hp) if hp is not None else None, check=True).items() if v is not None}
        ref = O.sample_events(events, coords, poses, q, K, hops=hp)
        for k in got:
            assert np.array_equal(got[k], ref[k]), k
    assert (ids[succ[q]] == ids[q]).all()
    empty = smp.sample_events(torch.empty((0,), dtype=torch.int64, device=DEV))
    assert empty["events_rays_start"].shape == (0, 3, 2)
    bad = events.copy()
    bad[q[0], -1] = float((q[0] + 1) % n if ids[(q[0] + 1) % n] != ids[q[0]] else (q[0] + 2) % n)      # a successor on another pixel
    if ids[int(bad[q[0], -1])] != ids[q[0]]:
        from evdeblurnerf_amd import _lib as L_
        with pytest.raises(L_.EvdError):
            EventSampler(bad, coords, poses, K).sample_events(T(q[:4]), check=True)


def test_precision_modes_under_larger_weights():
    """What the single-product float16 mode can and cannot promise.  The hidden-layer weights of the seed-derived network are
    scaled by 1.4 (activations and raw outputs grow ~10x, densities saturate more rays): the split-float16 mode must stay
    within 1e-4 of the exact-float32 kernel whatever the weights; float16 / bfloat16 degrade gracefully and are only reported."""
    from types import SimpleNamespace
    from evdeblurnerf_amd.renderer import NeRFAll
    sd = dict(W.prefixed(W.make_nerf_state_dict(23), "mlp_coarse"))
    for k in list(sd):
        if "pts_linears" in k and k.endswith("weight") and not k.endswith("pts_linears.0.weight"):
            sd[k] = (sd[k] * 1.4).astype(np.float32)
    args = SimpleNamespace(mode="nerf", netdepth=8, netwidth=256, multires=10, multires_views=4, use_viewdirs=True,
                           rgb_activate="sigmoid", sigma_activate="relu", N_importance=0)
    rays = T(W.synthetic_rays(17, 2048))
    kw = dict(ndc=True, near=0., far=1., use_viewdirs=True, N_samples=128, N_importance=0, retraw=False)
    out = {p: N(NeRFAll(args, sd, precision=p).eval().render(400, 400, W.synthetic_camera(), rays=rays, **kw)[0]) for p in ("f32", "f16x3", "f16", "bf16")}
    err = {p: maxabs(out[p], out["f32"]) for p in ("f16x3", "f16", "bf16")}
    print("RGB L-inf vs the f32 kernel with 1.4x hidden weights:", {k: f"{v:.2e}" for k, v in err.items()})
    assert err["f16x3"] < 1e-4
    assert err["f16"] < 2e-3 and err["bf16"] < 1e-1 and err["f16"] < err["bf16"]


def test_float16_modes_saturate_instead_of_overflowing():
    """Activations beyond the float16 range: the f16 kernels run with MODE.FP16_OVFL (conversions saturate at +-65504) and
    the split-float16 mode clamps at 65000, so a network whose first layer is scaled by 1e5 still yields finite outputs in
    every mode (and the exact-f32 kernel, which has no such limit, stays the reference for small activations elsewhere)."""
    from evdeblurnerf_amd.nerf import NeRF
    sd = W.make_nerf_state_dict(29)
    sd["pts_linears.0.weight"] = (sd["pts_linears.0.weight"] * 1e5).astype(np.float32)
    rs = np.random.RandomState(4)
    rb = np.zeros((512, 11), np.float32)
    rb[:, 3:6] = rs.uniform(-1, 1, (512, 3))
    rb[:, 7] = 1.0
    rb[:, 8:11] = [0.0, 0.6, 0.8]
    z = np.ones((512, 1), np.float32)
    for prec in ("f16", "f16x3", "bf16", "f32"):
        raw, _ = NeRF(sd, precision=prec).mlpforward(T(rb), T(z))
        assert torch.isfinite(raw).all(), prec


@pytest.mark.parametrize("S,white", [(128, False), (64, True), (192, False), (33, False)])
def test_raw2outputs_backward_matches_torch_autograd(S, white):
    """Backward of the compositing scan (evd_raw2outputs_bwd through NeRF.raw2outputs' autograd node) against torch autograd
    of a plain-torch restatement of nerf.py:74-129 (float64 on the GPU): d raw for random upstream gradients of
    (rgb_map, depth_map, acc_map, weights); relative L2 error <= 2e-5; saturated (alpha -> 1) and zero-density samples included."""
    from evdeblurnerf_amd.nerf import NeRF
    net = NeRF(W.make_nerf_state_dict(3))
    rs = np.random.RandomState(12 + S)
    R = 300
    raw_np = rs.standard_normal((R, S, 4)).astype(np.float32)
    raw_np[..., 3] *= rs.choice([0.0, 2.0, 40.0, 2000.0], size=(R, 1)).astype(np.float32)      # empty, thin, dense, opaque rays
    z_np = np.sort(rs.uniform(0, 1, (R, S)).astype(np.float32), -1)
    rd_np = rs.standard_normal((R, 3)).astype(np.float32)
    g = [torch.as_tensor(rs.standard_normal(sh).astype(np.float32), device=DEV) for sh in ((R, 3), (R,), (R,), (R, S))]

    def ref(raw, rd=None):
        raw, z, rd = raw.double(), T(z_np).double(), (T(rd_np) if rd is None else rd).double()
        rgb = torch.sigmoid(raw[..., :3])
        dists = (z[:, 1:] - z[:, :-1]) * rd.norm(dim=-1, keepdim=True)
        dens = torch.relu(raw[:, :-1, 3])
        alpha = torch.cat([1.0 - torch.exp(-dens * dists), torch.ones_like(z[:, :1])], -1)
        Tt = torch.cumprod(torch.cat([torch.ones_like(z[:, :1]), 1.0 - alpha + 1e-10], -1), -1)[:, :-1]
        w = alpha * Tt
        acc = w.sum(-1)
        rgb_map = (w[..., None] * rgb).sum(-2) + ((1.0 - acc[..., None]) if white else 0.0)
        return rgb_map, (w * z).sum(-1), acc, w

    raw_a = T(raw_np).requires_grad_(True)
    rd_a = T(rd_np).requires_grad_(True)          # the ray directions get a gradient too (evd_raw2outputs_bwd_rays: through dists * |d|)
    rgb, dens, acc, wts, depth, _ = net.raw2outputs(raw_a, T(z_np), rd_a, white_bkgd=white)
    (rgb * g[0]).sum().add((depth * g[1]).sum()).add((acc * g[2]).sum()).add((wts * g[3]).sum()).backward()
    raw_b = T(raw_np).requires_grad_(True)
    rd_b = T(rd_np).requires_grad_(True)
    r_rgb, r_depth, r_acc, r_w = ref(raw_b, rd_b)
    assert maxabs(N(rgb), N(r_rgb.float())) < 2e-6 and maxabs(N(wts), N(r_w.float())) < 2e-6
    ((r_rgb * g[0]).sum() + (r_depth * g[1]).sum() + (r_acc * g[2]).sum() + (r_w * g[3]).sum()).backward()
    ga, gb = raw_a.grad.double(), raw_b.grad.double()
    rel = float((ga - gb).norm() / gb.norm())
    print(f"[raw2outputs bwd S={S}] relative L2 error of d raw vs torch autograd (f64) = {rel:.2e}, L-inf {float((ga - gb).abs().max()):.2e}")
    assert rel < 2e-5
    assert float((ga - gb).abs().max()) < 1e-4 * max(1.0, float(gb.abs().max()))
    da, db = rd_a.grad.double(), rd_b.grad.double()
    rel_d = float((da - db).norm() / db.norm())
    print(f"[raw2outputs bwd S={S}] relative L2 error of d rays_d vs torch autograd (f64) = {rel_d:.2e}")
    assert rd_a.grad.shape == (R, 3) and rel_d < 2e-5


def test_sample_pdf_merge_properties_at_scale():
    """Size-independent properties of the fused sample_pdf + merge kernel at the full-frame size (160 000 rays, 64 + 128):
    the merged z are sorted, `order` is a permutation of the concatenation that reproduces them (index work: exact), new
    samples lie inside the coarse bin range, deterministic u gives identical results on a re-run, z_std is finite."""
    from evdeblurnerf_amd.rays import sample_pdf_merge
    rs = np.random.RandomState(21)
    R, S, Ni = 160_000, 64, 128
    z = T(np.sort(rs.uniform(0, 1, (R, S)).astype(np.float32), -1))
    w = T((rs.uniform(0, 1, (R, S)) ** 8).astype(np.float32))
    w[:100] = 0.0                                             # all-zero weights: uniform pdf (rays.py:152 adds 1e-5)
    zs, zm, order, zstd = sample_pdf_merge(z, w, Ni, det=True, want_order=True)
    assert zm.shape == (R, S + Ni) and order.shape == (R, S + Ni)
    assert bool((zm[:, 1:] >= zm[:, :-1]).all())
    cat = torch.cat([z, zs], -1)
    assert torch.equal(torch.gather(cat, 1, order.long()), zm)
    assert torch.equal(torch.sort(order.long(), -1).values, torch.arange(S + Ni, device=DEV).expand(R, -1))
    zmid_lo, zmid_hi = 0.5 * (z[:, :1] + z[:, 1:2]), 0.5 * (z[:, -2:-1] + z[:, -1:])
    assert bool((zs >= zmid_lo - 1e-6).all()) and bool((zs <= zmid_hi + 1e-6).all())
    zs2, zm2, order2, _ = sample_pdf_merge(z, w, Ni, det=True, want_order=True)
    assert torch.equal(zm, zm2) and torch.equal(order, order2)
    assert bool(torch.isfinite(zstd).all())


def test_training_through_the_fused_scan_reduces_the_loss():
    """The autograd node behind raw2outputs in an optimisation loop: a small PyTorch field (the caller's own network) is fitted
    to target colours THROUGH the fused compositing scan; the image loss must fall by 5x in 60 Adam steps."""
    from evdeblurnerf_amd.nerf import NeRF
    torch.manual_seed(0)
    net = NeRF(W.make_nerf_state_dict(3))
    R, S = 512, 64
    z = torch.linspace(0, 1, S, device=DEV).expand(R, S).contiguous()
    rd = torch.nn.functional.normalize(torch.randn(R, 3, device=DEV), dim=-1)
    pts = rd[:, None, :] * z[..., None]
    target = torch.rand(R, 3, device=DEV)
    field = torch.nn.Sequential(torch.nn.Linear(3, 64), torch.nn.ReLU(), torch.nn.Linear(64, 64), torch.nn.ReLU(), torch.nn.Linear(64, 4)).to(DEV)
    code = torch.nn.Parameter(torch.zeros(R, 1, 4, device=DEV))          # per-ray offset so the targets are reachable
    opt = torch.optim.Adam(list(field.parameters()) + [code], lr=2e-2)
    losses = []
    for _ in range(60):
        raw = field(pts) + code
        rgb = net.raw2outputs(raw, z, rd)[0]
        loss = ((rgb - target) ** 2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    print(f"loss through the fused scan: {losses[0]:.4f} -> {losses[-1]:.4f}")
    assert losses[-1] < 0.2 * losses[0]


@pytest.mark.parametrize("crf_kind,with_all", [("gamma", True), ("none", True), ("gamma", False)])
def test_blur_loss_backward_matches_torch_autograd(crf_kind, with_all):
    """Backward of the fused blur-loss reduction against torch autograd of a plain-torch restatement of run_nerf.py:443-497
    (float64): gradients of the assembled image loss w.r.t. the sub-exposure colours and both sets of composition weights."""
    from evdeblurnerf_amd.losses import blur_loss_partials_autograd, blur_loss_from_partials
    from evdeblurnerf_amd.tonemapping import CRF
    rs = np.random.RandomState(31)
    R, P = 257, 10
    mk = lambda *sh: torch.as_tensor(rs.uniform(0.05, 0.95, sh).astype(np.float32), device=DEV)
    rgb_p, rgb0_p, tgt, tgt0 = mk(R, P, 3), mk(R, P, 3), mk(R, 3), mk(R, 3)
    w1, w2 = torch.softmax(T(rs.standard_normal((R, P)).astype(np.float32)), -1), torch.softmax(T(rs.standard_normal((R, P)).astype(np.float32)), -1)
    crf = CRF(crf_kind)
    leaves = [t.clone().requires_grad_(True) for t in (rgb_p, rgb0_p, w1, w2)]
    kw = dict(rgb0_p=leaves[1], w2=leaves[3], target_pts0=tgt0) if with_all else {}
    p = blur_loss_partials_autograd(crf, leaves[0], leaves[2], tgt, **kw)
    loss, _ = blur_loss_from_partials(p, fine_loss_weight=0.3 if with_all else None, w_pts0=0.2)
    loss.backward()
    ref_leaves = [t.double().clone().requires_grad_(True) for t in (rgb_p, rgb0_p, w1, w2)]
    a, b, w1d, w2d = ref_leaves
    f = (lambda x: x ** (1.0 / 2.2)) if crf_kind == "gamma" else (lambda x: x)
    mse = lambda x, y: ((f(x) - y.double()) ** 2).mean()
    rgb = (a * w1d[..., None]).sum(1)
    if with_all:
        img = mse(rgb, tgt) + mse((b * w1d[..., None]).sum(1), tgt)
        fine = mse((a * w2d[..., None]).sum(1), tgt)
        ref = img * 0.7 + fine * 0.3 + 0.2 * (mse(a[:, 0], tgt0) + mse(b[:, 0], tgt0))
    else:
        ref = mse(rgb, tgt)
    assert abs(float(loss.detach()) - float(ref.detach())) < 1e-6
    ref.backward()
    for name, x, y in zip(("rgb_p", "rgb0_p", "w1", "w2"), leaves, ref_leaves):
        if y.grad is None:
            assert x.grad is None or float(x.grad.abs().max()) == 0.0, name
            continue
        rel = float((x.grad.double() - y.grad).norm() / y.grad.norm())
        assert rel < 2e-5, (name, rel)


def test_loss_backward_entries_host_and_device_gradients_agree():
    """evd_blur_loss_bwd / evd_event_loss_bwd take dL/d partial as host scalars, the *_dev entries read the same values from device memory
    (what the autograd nodes call: no host copy inside a backward pass): the same outputs."""
    import ctypes as C
    from evdeblurnerf_amd import _lib as L
    from evdeblurnerf_amd.tonemapping import CRF
    rs = np.random.RandomState(77)
    R, P, n = 129, 10, 93
    mk = lambda *sh: T(rs.uniform(0.05, 0.95, sh).astype(np.float32))
    rgb_p, rgb0_p, tgt, tgt0 = mk(R, P, 3), mk(R, P, 3), mk(R, 3), mk(R, 3)
    w1, w2 = torch.softmax(mk(R, P), -1).contiguous(), torch.softmax(mk(R, P), -1).contiguous()
    g = np.array([0.31, -0.12, 0.55, 0.07, 0.21, 0.0, 0.0, 0.0], np.float32)
    crf = CRF("gamma")
    outs = []
    for dev in (False, True):
        d = [torch.empty_like(rgb_p), torch.empty_like(rgb0_p), torch.empty_like(w1), torch.empty_like(w2)]
        gd = T(g)
        fn = L.lib().evd_blur_loss_bwd_dev if dev else L.lib().evd_blur_loss_bwd
        garg = L.ptr(gd) if dev else g.ctypes.data_as(C.POINTER(C.c_float))
        L.check(fn(crf.handle, 0, L.ptr(rgb_p), L.ptr(rgb0_p), L.ptr(w1), L.ptr(w2), L.ptr(tgt), L.ptr(tgt0), R, P, garg, L.ptr(d[0]), L.ptr(d[1]),
                   L.ptr(d[2]), L.ptr(d[3]), L.stream_ptr()), "evd_blur_loss_bwd")
        outs.append([N(t) for t in d])
    for k, (a, b) in enumerate(zip(*outs)):
        # d rgb: one lane each, bit for bit; d w1 / d w2: three channels summed by float atomics -- equal to rounding
        assert np.array_equal(a, b) if k < 2 else np.abs(a - b).max() <= 1e-6 * np.abs(a).max()
    sd = {k: (v * (3.0 if "weight" in k else 1.0)).astype(np.float32) for k, v in W.make_crf_state_dict(51, 2).items()}
    crf_ev = CRF("learn", state_dict=sd, extra_features=2)
    es, ee, es0, ee0 = mk(n, 3), mk(n, 3), mk(n, 3), mk(n, 3)
    cn, cp = T(-rs.randint(0, 4, n).astype(np.float32)), T(rs.randint(0, 4, n).astype(np.float32))
    npar = int(L.lib().evd_crf_param_count())
    outs = []
    for dev in (False, True):
        d = [torch.zeros_like(es) for _ in range(4)] + [torch.empty((npar,), device=DEV)]
        gd = T(g[:3])
        if dev:
            L.check(L.lib().evd_event_loss_bwd_dev(crf_ev.handle, 0, 1, 0, L.ptr(es), L.ptr(ee), L.ptr(es0), L.ptr(ee0), L.ptr(cn), L.ptr(cp), 0.2, 0.2, None, None,
                                                   n, L.ptr(gd), L.ptr(d[0]), L.ptr(d[1]), L.ptr(d[2]), L.ptr(d[3]), L.ptr(d[4]), L.stream_ptr()), "evd_event_loss_bwd_dev")
        else:
            L.check(L.lib().evd_event_loss_bwd(crf_ev.handle, 0, 1, 0, L.ptr(es), L.ptr(ee), L.ptr(es0), L.ptr(ee0), L.ptr(cn), L.ptr(cp), 0.2, 0.2, None, None,
                                               n, float(g[0]), float(g[1]), L.ptr(d[0]), L.ptr(d[1]), L.ptr(d[2]), L.ptr(d[3]), L.ptr(d[4]), L.stream_ptr()), "evd_event_loss_bwd")
        outs.append([N(t) for t in d])
    for a, b in zip(outs[0][:4], outs[1][:4]):
        assert np.array_equal(a, b)
    # (the CRF parameter gradient is a float atomic sum over the blocks: equal to rounding, not bit for bit)
    assert np.abs(outs[0][4] - outs[1][4]).max() <= 1e-5 * max(1.0, np.abs(outs[0][4]).max())


@pytest.mark.parametrize("cfg", ["blender", "cdavis"])
def test_event_loss_backward_matches_torch_autograd(cfg):
    """Backward of the fused event-loss reduction (learnable event-CRF included) against torch autograd of a plain-torch
    restatement of tonemapping.py:59-93 + run_nerf.py:518-570 + events.py:260-284 (float64): gradients w.r.t. the four colour
    inputs and w.r.t. every CRF parameter.  'blender': pos-neg features + rec601 luma; 'cdavis': colour mask, per-colour
    features and weights, tonemap_only."""
    from evdeblurnerf_amd.losses import event_loss_partials_autograd, event_loss_from_partials, crf_param_grads
    from evdeblurnerf_amd.tonemapping import CRF
    rs = np.random.RandomState(41)
    n = 277
    sd = W.make_crf_state_dict(51, 2)
    sd = {k: (v * (3.0 if "weight" in k else 1.0)).astype(np.float32) for k, v in sd.items()}      # not-near-identity CRF: real gradients everywhere
    crf = CRF("learn", state_dict=sd, extra_features=2)
    mk = lambda: T(rs.uniform(0.05, 0.95, (n, 3)).astype(np.float32))
    es, ee, es0, ee0 = mk(), mk(), mk(), mk()
    cn, cp = T(-rs.randint(0, 4, n).astype(np.float32)), T(rs.randint(0, 4, n).astype(np.float32))
    thr = 0.2 if cfg == "blender" else 0.25
    cmask = np.zeros((n, 3), np.uint8)
    cmask[np.arange(n), rs.randint(0, 3, n)] = 1
    cwt = [0.4, 0.2, 0.4]
    kw = dict(add_bii="pos-neg") if cfg == "blender" else dict(add_bii="color-pos-neg", tonemap_only=True, color_mask=T(cmask), color_weight=cwt)
    leaves = [t.clone().requires_grad_(True) for t in (es, ee, es0, ee0)]
    theta = torch.zeros(705, device=DEV, requires_grad=True)
    p = event_loss_partials_autograd(crf, theta, leaves[0], leaves[1], cn, cp, thr, thr, start0=leaves[2], end0=leaves[3], **kw)
    loss = event_loss_from_partials(p)
    loss.backward()
    # ---- plain-torch float64 restatement
    P = {k: torch.as_tensor(v, device=DEV).double().requires_grad_(True) for k, v in sd.items()}
    ref_leaves = [t.double().clone().requires_grad_(True) for t in (es, ee, es0, ee0)]

    def crf_t(x, feat):                       # x [n,3], feat [n,3,2]
        xin = torch.cat([x.reshape(-1, 1), feat.reshape(-1, 2)], -1)
        h = torch.relu(xin @ P["linear.0.weight"].T + P["linear.0.bias"])
        h = torch.relu(h @ P["linear.2.weight"].T + P["linear.2.bias"])
        h = torch.relu(h @ P["linear.4.weight"].T + P["linear.4.bias"])
        res = (h @ P["linear.6.weight"].T + P["linear.6.bias"]) * 0.1
        return torch.sigmoid(res + x.reshape(-1, 1)).reshape(x.shape)

    f2 = torch.stack([cn, cp], -1).double()
    cm_t = torch.as_tensor(cmask.astype(bool), device=DEV)
    if cfg == "blender":
        feat = f2[:, None, :].expand(n, 3, 2)
        lum = lambda x: (crf_t(x, feat) * torch.tensor([0.299, 0.587, 0.114], device=DEV, dtype=torch.float64)).sum(-1)
        wgt = torch.ones(n, device=DEV, dtype=torch.float64)
    else:
        feat = f2[:, None, :] * cm_t[..., None].double()
        lum = lambda x: crf_t(x, feat)[cm_t]
        wgt = torch.tensor(cwt, device=DEV, dtype=torch.float64)[cm_t.double().argmax(-1)]
    bii = thr * cn.double() + thr * cp.double()
    egm = lambda a, b: (wgt * (torch.log(lum(b) + 1e-5) - torch.log(lum(a) + 1e-5) - bii) ** 2).sum() / wgt.sum()
    ref = egm(ref_leaves[2], ref_leaves[3]) + egm(ref_leaves[0], ref_leaves[1])
    assert abs(float(loss.detach()) - float(ref.detach())) < 2e-5 * max(1.0, float(ref.detach()))
    ref.backward()
    for name, x, y in zip(("start", "end", "start0", "end0"), leaves, ref_leaves):
        rel = float((x.grad.double() - y.grad).norm() / y.grad.norm())
        assert rel < 5e-5, (name, rel)
    got = crf_param_grads(theta.grad, 2)
    for k, v in P.items():
        rel = float((got[k].double() - v.grad).norm() / max(float(v.grad.norm()), 1e-12))
        assert rel < 2e-4, (k, rel)


def test_device_side_numerics_flags(capsys):
    """renderer.py:259-263 (isnan / isinf per key, two host syncs each) as one launch into a device flag word per key
    (evd_numerics_flags): exact against torch on arrays with planted NaN / Inf (aligned, unaligned, empty), through
    render_rays without a sync (check_numerics="device"), and with the reference's printed messages (check_numerics=True)."""
    from types import SimpleNamespace
    from evdeblurnerf_amd.renderer import NeRFAll
    sd = dict(W.prefixed(W.make_nerf_state_dict(11), "mlp_coarse"))
    args = SimpleNamespace(mode="nerf", netdepth=8, netwidth=256, multires=10, multires_views=4, use_viewdirs=True,
                           rgb_activate="sigmoid", sigma_activate="relu", N_importance=0)
    model = NeRFAll(args, sd, precision="f16x3").eval()
    rs = np.random.RandomState(0)
    a = rs.standard_normal(100003).astype(np.float32)
    b = a.copy(); b[77777] = np.nan
    c = a.copy(); c[100002] = np.inf; c[5] = -np.inf
    d = a.copy(); d[0] = np.nan; d[99] = np.inf
    tens = [T(a), T(b), T(c), T(d), T(b)[1:], T(c)[100000:], torch.empty((0,), device=DEV), T(a)[3:50001]]
    got = model.numerics_flags(tens).tolist()
    want = [int(torch.isnan(t).any()) | 2 * int(torch.isinf(t).any()) for t in tens]
    assert got == want == [0, 1, 2, 3, 1, 2, 0, 0]
    rays = T(W.synthetic_rays(3, 200))
    kw = dict(ndc=True, near=0., far=1., use_viewdirs=True, N_samples=32, N_importance=0, retraw=True)
    ex = model.render(400, 400, W.synthetic_camera(), rays=rays, check_numerics="device", chunk=64, **kw)[3]
    assert ex["numerics_flags"].is_cuda and ex["numerics_flags"].tolist() == [0] * len(ex["numerics_keys"])
    assert {"rgb_map", "depth_map", "acc_map", "weights", "z_vals"} <= set(ex["numerics_keys"])
    bad = dict(sd)
    bad["mlp_coarse.rgb_linear.bias"] = np.array([np.nan, 0.0, 0.0], np.float32)
    mb = NeRFAll(args, bad, precision="f32").eval()
    ex = mb.render(400, 400, W.synthetic_camera(), rays=rays, check_numerics="device", **kw)[3]
    flags = dict(zip(ex["numerics_keys"], ex["numerics_flags"].tolist()))
    assert flags["rgb_map"] & 1 and flags["acc_map"] == 0 and flags["z_vals"] == 0
    mb.render(400, 400, W.synthetic_camera(), rays=rays, check_numerics=True, **kw)
    assert "! [Numerical Error] rgb_map contains nan." in capsys.readouterr().out


def test_crf_init_identity():
    """CRF.init_identity (tonemapping.py:29-57, `tone_mapping_learn_init_identity` of the shipped configs): after the 3000-step
    pre-training the learnable CRF -- evaluated by the library's own kernel -- is close to the identity on (0, 1) with the extra
    features at zero, deterministic for a seed, and its parameters round-trip through the handle."""
    from evdeblurnerf_amd.tonemapping import TonemappingTransform
    tm = TonemappingTransform("gamma", "learn", init_learn_identity=True, extra_features_event=2)
    x = torch.rand((4096, 3), device=DEV) * 0.9 + 0.05
    y = tm.tonemapping_event(x, x_feat=torch.zeros((4096, 2), device=DEV))
    err = (y - x).abs()
    print(f"CRF init_identity: mean |crf(x) - x| = {float(err.mean()):.4f}, max = {float(err.max()):.4f}")
    # sigmoid(0.1 mlp + x) = x needs mlp = 10 (logit(x) - x): a 16-wide network gets within a few hundredths (measured 0.019 / 0.048)
    assert float(err.mean()) < 0.03 and float(err.max()) < 0.08
    sd1 = tm.tonemapping_event.identity_state_dict(2, 42, steps=50)
    sd2 = tm.tonemapping_event.identity_state_dict(2, 42, steps=50)
    assert all(np.array_equal(sd1[k], sd2[k]) for k in sd1) and set(sd1) == {f"linear.{i}.{k}" for i in (0, 2, 4, 6) for k in ("weight", "bias")}
    with pytest.raises(Exception):
        TonemappingTransform("gamma", "learn")               # neither parameters nor init_learn_identity
