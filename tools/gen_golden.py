#!/usr/bin/env python
"""Generate tests/golden/*.npz by RUNNING the imported reference on CPU.

Run in the build container only (needs /root/reference):

    python tools/gen_golden.py            # all cases
    python tools/gen_golden.py G3 G5      # a subset

Each fixture holds inputs that are not seed-derivable plus the reference's
outputs (float32). Parameters are never stored: both sides derive them from
``evdeblurnerf_amd.weights`` (numpy RandomState) and this script
``load_state_dict``s them into the reference modules. No reference source is
copied; this script only calls it. Golden ids follow SURVEY.md 8c.
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_import  # noqa: E402

ref_import.install()

import torch  # noqa: E402

from evdeblurnerf_amd import weights as W  # noqa: E402

torch.manual_seed(0)
torch.set_grad_enabled(False)
OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def n(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    arrays = {k: (np.asarray(v, dtype=np.float32) if np.asarray(v).dtype == np.float64 else np.asarray(v))
              for k, v in arrays.items()}
    np.savez_compressed(path, **arrays)
    print(f"  wrote {os.path.relpath(path, ROOT)}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def save64(name, **arrays):
    """like save(), keeping float64 arrays (timestamps in microseconds do not fit float32)"""
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"  wrote {os.path.relpath(path, ROOT)}  ({os.path.getsize(path) / 1024:.1f} KiB)")


# ---------------------------------------------------------------------------
def G1_embedder():
    from networks.embedding import get_embedder
    rs = np.random.RandomState(101)
    x = rs.uniform(-1.5, 1.5, size=(192, 3)).astype(np.float32)
    x[:8] = rs.uniform(-40.0, 40.0, size=(8, 3)).astype(np.float32)  # large magnitudes
    x[8] = 0.0
    x[9] = [1.0, -1.0, 0.5]
    e10, d10 = get_embedder(10)
    e4, d4 = get_embedder(4)
    e2, d2 = get_embedder(2)
    assert d10 == 63 and d4 == 27
    save("G1_embedder", x=x, pe10=n(e10(t(x))), pe4=n(e4(t(x))), pe2=n(e2(t(x))))


def _ref_nerf(seed, D=8, Wd=256, rgb_add_bias=True, **kw):
    from networks.nerf import NeRF
    net = NeRF(D=D, W=Wd, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True,
               rgb_add_bias=rgb_add_bias, **kw)
    sd = W.make_nerf_state_dict(seed, D=D, W=Wd, rgb_add_bias=rgb_add_bias)
    ref_import.load_np_state_dict(net, sd)
    net.train(False)
    return net


def G2_nerf_mlp():
    rs = np.random.RandomState(202)
    pts = rs.uniform(-1.2, 1.2, size=(384, 3)).astype(np.float32)
    dirs = rs.standard_normal((384, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    from networks.embedding import get_embedder
    e10, _ = get_embedder(10)
    e4, _ = get_embedder(4)
    x = torch.cat([e10(t(pts)), e4(t(dirs))], -1)
    out = {}
    for tag, kw in (("w256", dict(seed=7, D=8, Wd=256, rgb_add_bias=True)),
                    ("w256_nobias", dict(seed=8, D=8, Wd=256, rgb_add_bias=False)),
                    ("w64", dict(seed=9, D=8, Wd=64, rgb_add_bias=True))):
        net = _ref_nerf(**kw)
        raw, feat = net.eval(x)
        out[f"raw_{tag}"] = n(raw)
        out[f"feat_{tag}"] = n(feat)[:, :16]  # first 16 feature columns are enough to pin the layer
        net_b = _ref_nerf(**kw, extract_feature="before_linear")
        _, featb = net_b.eval(x)
        out[f"featb_{tag}"] = n(featb)[:, :16]
    save("G2_nerf_mlp", pts=pts, dirs=dirs, **out)


def _composite_inputs(rs, R, S):
    raw = rs.standard_normal((R, S, 4)).astype(np.float32)
    raw[..., 3] = raw[..., 3] * 4.0 + 1.0
    near = rs.uniform(0.0, 0.1, size=(R, 1)).astype(np.float32)
    z = np.sort(near + rs.uniform(0.0, 1.0, size=(R, S)).astype(np.float32) * (1.0 - near), axis=-1)
    rays_d = rs.standard_normal((R, 3)).astype(np.float32)
    # edge cases per SURVEY 8c G3
    raw[0, :, 3] = -5.0                     # zero density everywhere (relu clamps)
    raw[1, :, 3] = 1.0e4                    # huge density: first sample takes everything
    z[2, 10:20] = z[2, 10]                  # duplicate z => delta 0 => alpha 0
    raw[3, :, 3] = 0.0                      # exactly zero
    z[4] = np.linspace(0.0, 1.0, S)         # regular spacing
    raw[5, : S // 2, 3] = -1.0
    raw[5, S // 2:, 3] = 50.0               # wall in the middle
    return raw, z.astype(np.float32), rays_d


def G3_nerf_raw2outputs():
    from networks.nerf import NeRF
    rs = np.random.RandomState(303)
    out = {}
    for S in (64, 128, 33):
        raw, z, rays_d = _composite_inputs(rs, 20, S)
        out[f"raw_S{S}"], out[f"z_S{S}"], out[f"d_S{S}"] = raw, z, rays_d
        for tag, kw, call in (
                ("plain", {}, {}),
                ("white", {}, {"white_bkgd": True}),
                ("rmnear", {"render_rmnearplane": 20}, {}),
                ("relu_rgb", {"rgb_activate": "relu"}, {}),
                ("none_rgb", {"rgb_activate": "none"}, {}),
                ("softplus", {"sigma_activate": "softplus"}, {})):
            net = NeRF(D=2, W=8, input_ch=3, input_ch_views=3, use_viewdirs=True, **kw)
            net.train(False)
            rgb, dens, acc, wts, depth, fmap = net.raw2outputs(t(raw), t(z), t(rays_d), None, 0, **call)
            assert fmap is None
            out[f"rgb_S{S}_{tag}"] = n(rgb)
            out[f"acc_S{S}_{tag}"] = n(acc)
            out[f"depth_S{S}_{tag}"] = n(depth)
            out[f"weights_S{S}_{tag}"] = n(wts)
            if tag == "plain":
                out[f"density_S{S}"] = n(dens)
        # feature compositing (composite_feature=True path, nerf.py:119)
        feat = rs.standard_normal((20, S, 5)).astype(np.float32)
        net = NeRF(D=2, W=8, input_ch=3, input_ch_views=3, use_viewdirs=True)
        net.train(False)
        fmap = net.raw2outputs(t(raw), t(z), t(rays_d), t(feat), 0)[5]
        out[f"feat_S{S}"], out[f"fmap_S{S}"] = feat, n(fmap)
    save("G3_nerf_raw2outputs", **out)


def G4_voxel_raw2outputs():
    from networks.pdrf.voxnerf import VoxelNeRFRayFeatures, VoxelNeRFSampleFeatures
    rs = np.random.RandomState(404)
    aabb = (torch.tensor([-1.5, -1.5, -1.0]), torch.tensor([1.5, 1.5, 1.0]))
    out = {}
    for S in (64, 128):
        raw4, z, rays_d = _composite_inputs(rs, 12, S)
        # voxel order: sigma first, rgb after (voxnerf.py:172,179); colours arrive already sigmoided
        raw = np.concatenate([raw4[..., 3:4], 1.0 / (1.0 + np.exp(-raw4[..., :3]))], -1).astype(np.float32)
        out[f"raw_S{S}"], out[f"z_S{S}"], out[f"d_S{S}"] = raw, z, rays_d
        for tag, cls in (("coarse", VoxelNeRFRayFeatures), ("fine", VoxelNeRFSampleFeatures)):
            net = cls(aabb=aabb, input_ch=95, n_voxels=512, app_n_comp=[4, 2, 2], app_dim=4)
            rgb, dens, acc, wts, depth = net.raw2outputs(t(raw), t(z), t(rays_d), 0, is_train=False)
            out[f"rgb_S{S}_{tag}"], out[f"acc_S{S}_{tag}"] = n(rgb), n(acc)
            out[f"depth_S{S}_{tag}"], out[f"weights_S{S}_{tag}"] = n(depth), n(wts)
        # 16-channel feature compositing (PBE mode, voxnerf.py:226)
        raw16 = rs.standard_normal((12, S, 16)).astype(np.float32)
        net = VoxelNeRFRayFeatures(aabb=aabb, input_ch=95, n_voxels=512, app_n_comp=[4, 2, 2], app_dim=4)
        fm, _, acc, wts, depth = net.raw2outputs(t(raw16), t(z), t(rays_d), 0, is_train=False)
        out[f"raw16_S{S}"], out[f"fmap16_S{S}"] = raw16, n(fm)
    save("G4_voxel_raw2outputs", **out)


def G5_sample_pdf():
    from utils.rays import sample_pdf
    rs = np.random.RandomState(505)
    out = {}
    for S, N in ((64, 64), (64, 128), (128, 64), (17, 9)):
        R = 40
        z = np.sort(rs.uniform(0, 1, size=(R, S)).astype(np.float32), -1)
        z[0] = np.linspace(0, 1, S)
        bins = (0.5 * (z[:, 1:] + z[:, :-1])).astype(np.float32)
        w = rs.uniform(0, 1, size=(R, S - 2)).astype(np.float32) ** 4
        w[0] = 0.0                          # all-zero weights => uniform pdf
        w[1] = 0.0
        w[1, (S - 2) // 3] = 1.0            # one spike
        w[2] = 1.0                          # flat
        w[3] = 0.0
        w[3, 0] = 1.0                       # spike in first bin
        w[4] = 0.0
        w[4, -1] = 1.0                      # spike in last bin
        w[5] = 1e-7 * rs.uniform(0, 1, size=S - 2)  # tiny weights: denom guard
        zs = sample_pdf(t(bins), t(w), N, det=True)
        u = rs.uniform(0, 1, size=(R, N)).astype(np.float32)
        # explicit-u variant: patch torch.rand for one call (the reference draws u internally)
        orig = torch.rand
        torch.rand = lambda *a, **k: t(u)
        try:
            zs_u = sample_pdf(t(bins), t(w), N, det=False)
        finally:
            torch.rand = orig
        key = f"S{S}_N{N}"
        out[f"bins_{key}"], out[f"w_{key}"], out[f"u_{key}"] = bins, w, u
        out[f"det_{key}"], out[f"rand_{key}"] = n(zs), n(zs_u)
    save("G5_sample_pdf", **out)


def G6_rays():
    from utils.rays import get_rays, get_rays_pix, get_ndc_rays
    K = W.synthetic_camera(400, 400, 400.0)
    Kn = W.synthetic_camera(60, 80, 95.5)
    Kn[0, 2] += 1.25
    Kn[1, 2] -= 0.75
    c2w = W.synthetic_pose(5)
    o, d = get_rays(60, 80, t(Kn), t(c2w))
    rs = np.random.RandomState(606)
    coords = np.stack([rs.randint(0, 400, 300), rs.randint(0, 400, 300)], -1).astype(np.float32)
    poses = np.stack([W.synthetic_pose(60 + i) for i in range(300)])
    op, dp = get_rays_pix(t(coords), t(K), t(poses))
    on, dn = get_ndc_rays(400, 400, float(K[0, 0]), 1.0, op, dp)
    save("G6_rays", Kn=Kn, c2w=c2w, rays_o_full=n(o)[::7, ::5], rays_d_full=n(d)[::7, ::5],
         coords=coords, poses=poses, rays_o_pix=n(op), rays_d_pix=n(dp), ndc_o=n(on), ndc_d=n(dn))


def _nerfall(mode, N_importance, seed, **over):
    from networks.renderer import NeRFAll
    over.setdefault("rgb_add_bias", True)
    args = ref_import.blurfactory_args(mode=mode, N_importance=N_importance, **over)
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        model = NeRFAll(args)
    return model, args


def G7_render_nerf():
    K = W.synthetic_camera()
    out = {}
    # (a) hierarchical 64+64, retraw
    model, _ = _nerfall("nerf", 64, 0)
    sd = {}
    sd.update(W.prefixed(W.make_nerf_state_dict(11), "mlp_coarse"))
    sd.update(W.prefixed(W.make_nerf_state_dict(12), "mlp_fine"))
    ref_import.load_np_state_dict(model, sd)
    model.train(False)
    rays = W.synthetic_rays(1, 96)
    rgb, depth, acc, ex = model.render(400, 400, t(K), 1024, rays=t(rays), ndc=True, near=0., far=1.,
                                       use_viewdirs=True, N_samples=64, N_importance=64, retraw=True,
                                       perturb=0., raw_noise_std=0.)
    out.update(a_rgb=n(rgb), a_depth=n(depth), a_acc=n(acc), **{f"a_{k}": n(v) for k, v in ex.items()})
    # (b) the metric shape at small R: 128 samples, single pass, chunked (chunk < R)
    model, _ = _nerfall("nerf", 0, 0)
    ref_import.load_np_state_dict(model, W.prefixed(W.make_nerf_state_dict(13), "mlp_coarse"))
    model.train(False)
    rays = W.synthetic_rays(2, 80)
    rgb, depth, acc, ex = model.render(400, 400, t(K), 32, rays=t(rays), ndc=True, near=0., far=1.,
                                       use_viewdirs=True, N_samples=128, N_importance=0, retraw=True,
                                       perturb=0., raw_noise_std=0.)
    out.update(b_rgb=n(rgb), b_depth=n(depth), b_acc=n(acc), **{f"b_{k}": n(v) for k, v in ex.items()})
    # (c) config 1: R=1024-ray batch shape scaled down, 64 samples, white background, no ndc
    rays = W.synthetic_rays(3, 64)
    rgb, depth, acc, ex = model.render(400, 400, t(K), 1024, rays=t(rays), ndc=False, near=0.5, far=3.5,
                                       use_viewdirs=True, N_samples=64, N_importance=0, retraw=False,
                                       perturb=0., raw_noise_std=0., white_bkgd=True, lindisp=True)
    out.update(c_rgb=n(rgb), c_depth=n(depth), c_acc=n(acc))
    # (d) explicit randomness: perturb=1 with injected t_rand / u (stratified + sample_pdf draws)
    model, _ = _nerfall("nerf", 32, 0)
    sd = {}
    sd.update(W.prefixed(W.make_nerf_state_dict(11), "mlp_coarse"))
    sd.update(W.prefixed(W.make_nerf_state_dict(12), "mlp_fine"))
    ref_import.load_np_state_dict(model, sd)
    model.train(False)
    rays = W.synthetic_rays(4, 48)
    rs = np.random.RandomState(707)
    t_rand = rs.uniform(0, 1, size=(48, 64)).astype(np.float32)
    u = rs.uniform(0, 1, size=(48, 32)).astype(np.float32)
    draws = [t(t_rand), t(u)]
    orig = torch.rand
    torch.rand = lambda *a, **k: draws.pop(0)
    try:
        rgb, depth, acc, ex = model.render(400, 400, t(K), 1024, rays=t(rays), ndc=True, near=0., far=1.,
                                           use_viewdirs=True, N_samples=64, N_importance=32, retraw=True,
                                           perturb=1., raw_noise_std=0.)
    finally:
        torch.rand = orig
    assert not draws
    out.update(d_t_rand=t_rand, d_u=u, d_rgb=n(rgb), d_depth=n(depth), d_acc=n(acc),
               **{f"d_{k}": n(v) for k, v in ex.items()})
    save("G7_render_nerf", **out)


def G23_render_nerf_no_viewdirs():
    """mode='nerf' with use_viewdirs=False (renderer.py:443-446: 8-column ray batch; nerf.py:41-44,158-160: output_linear head).  In the
    reference this network only runs with extract_feature "before_linear" (nerf.py:159 asserts otherwise), i.e. with kernel_use_awp and
    an awpnet object; output_ch is 5 with importance sampling (renderer.py:46) and raw2outputs reads channels 0..3."""
    from networks.renderer import NeRFAll
    import contextlib
    import io
    K = W.synthetic_camera()
    out = {}
    for tag, Ni, S, R, ndc in (("a", 32, 48, 72, True), ("b", 0, 128, 40, False)):
        args = ref_import.blurfactory_args(mode="nerf", N_importance=Ni, use_viewdirs=False, kernel_use_awp=True, rgb_add_bias=True)
        with contextlib.redirect_stdout(io.StringIO()):
            model = NeRFAll(args, awpnet=object())
        assert model.extract_feature == "before_linear" and model.output_ch == (5 if Ni > 0 else 4)
        och = model.output_ch
        sd = dict(W.prefixed(W.make_nerf_state_dict(61, input_ch_views=0, use_viewdirs=False, output_ch=och), "mlp_coarse"))
        if Ni > 0:
            sd.update(W.prefixed(W.make_nerf_state_dict(62, input_ch_views=0, use_viewdirs=False, output_ch=och), "mlp_fine"))
        ref_import.load_np_state_dict(model, sd)
        model.train(False)
        rays = W.synthetic_rays(23 + Ni, R)
        kw = dict(ndc=ndc, near=0. if ndc else 0.5, far=1. if ndc else 3.5, use_viewdirs=False, N_samples=S, N_importance=Ni, retraw=True,
                  perturb=0., raw_noise_std=0., inference=True)
        rgb, depth, acc, ex = model.render(400, 400, t(K), 1 << 20, rays=t(rays), **kw)
        out.update({f"{tag}_rgb": n(rgb), f"{tag}_depth": n(depth), f"{tag}_acc": n(acc)},
                   **{f"{tag}_{k}": n(v) for k, v in ex.items() if isinstance(v, torch.Tensor)})
        kw["inference"] = False                 # with the per-sample "before_linear" feature AWP consumes (renderer.py:253-256)
        _, _, _, ex = model.render(400, 400, t(K), 1 << 20, rays=t(rays[:2]), **kw)
        out[f"{tag}_depth_feature2"] = n(ex["depth_feature"])
    save("G23_render_nerf_no_viewdirs", **out)


PDRF_SMALL = dict(coarse_n_voxels=24 ** 3, fine_n_voxels=48 ** 3)


def _pdrf_sds(model, seed_c, seed_f):
    gc = [int(v) for v in model.mlp_coarse.gridSize]
    sd = W.prefixed(W.make_pdrf_state_dict(seed_c, gc, input_ch=32 + 63, hidden_dim=64, geo_feat_dim=15),
                    "mlp_coarse")
    if model.mlp_fine is not None:
        gf = [int(v) for v in model.mlp_fine.gridSize]
        sd.update(W.prefixed(W.make_pdrf_state_dict(seed_f, gf, input_ch=64 + 63, hidden_dim=256,
                                                    geo_feat_dim=128), "mlp_fine"))
    return sd


def G24_render_other_multires():
    """frequency counts other than the default (options.py:94-97 --multires 10 --multires_views 4): embedding.py:101-117 builds the
    encoders from them, renderer.py:18,44 sizes the networks' inputs."""
    K = W.synthetic_camera()
    out = {}
    # (a) mode='nerf', multires 6 / multires_views 2, hierarchical 48 + 32
    L, Lv = 6, 2
    model, _ = _nerfall("nerf", 32, 0, multires=L, multires_views=Lv)
    sd = {}
    sd.update(W.prefixed(W.make_nerf_state_dict(71, input_ch=W.pe_dim(L), input_ch_views=W.pe_dim(Lv)), "mlp_coarse"))
    sd.update(W.prefixed(W.make_nerf_state_dict(72, input_ch=W.pe_dim(L), input_ch_views=W.pe_dim(Lv)), "mlp_fine"))
    ref_import.load_np_state_dict(model, sd)
    model.train(False)
    rays = W.synthetic_rays(41, 56)
    rgb, depth, acc, ex = model.render(400, 400, t(K), 1024, rays=t(rays), ndc=True, near=0., far=1., use_viewdirs=True, N_samples=48,
                                       N_importance=32, retraw=True, perturb=0., raw_noise_std=0.)
    out.update(a_rgb=n(rgb), a_depth=n(depth), a_acc=n(acc), **{f"a_{k}": n(v) for k, v in ex.items()})
    # (b) mode='nerf', multires 3 / multires_views 8 (direction encoding wider than the default's two k-steps), 4 x 64 network, skip at 1
    L, Lv = 3, 8
    model, _ = _nerfall("nerf", 0, 0, multires=L, multires_views=Lv, netdepth=4, netwidth=64)
    # (skips = [4] is fixed in renderer.py:48: with D = 4 it never fires, nerf.py:24,137)
    ref_import.load_np_state_dict(model, W.prefixed(W.make_nerf_state_dict(73, D=4, W=64, input_ch=W.pe_dim(L), input_ch_views=W.pe_dim(Lv), skips=()),
                                                    "mlp_coarse"))
    model.train(False)
    rays = W.synthetic_rays(42, 40)
    rgb, depth, acc, ex = model.render(400, 400, t(K), 1024, rays=t(rays), ndc=False, near=0.5, far=3.5, use_viewdirs=True, N_samples=64,
                                       N_importance=0, retraw=True, perturb=0., raw_noise_std=0.)
    out.update(b_rgb=n(rgb), b_depth=n(depth), b_acc=n(acc), **{f"b_{k}": n(v) for k, v in ex.items()})
    # (c) mode='c2f', multires 7 / multires_views 3 on the small grids of G9
    L, Lv = 7, 3
    model, _ = _nerfall("c2f", 32, 0, rgb_add_bias=False, multires=L, multires_views=Lv, **PDRF_SMALL)
    gc = [int(v) for v in model.mlp_coarse.gridSize]
    gf = [int(v) for v in model.mlp_fine.gridSize]
    sd = W.prefixed(W.make_pdrf_state_dict(74, gc, input_ch=32 + W.pe_dim(L), input_ch_views=W.pe_dim(Lv), hidden_dim=64, geo_feat_dim=15), "mlp_coarse")
    sd.update(W.prefixed(W.make_pdrf_state_dict(75, gf, input_ch=64 + W.pe_dim(L), input_ch_views=W.pe_dim(Lv), hidden_dim=256, geo_feat_dim=128), "mlp_fine"))
    ref_import.load_np_state_dict(model, sd)
    model.train(False)
    rays = W.synthetic_rays(43, 48)
    rgb, depth, acc, ex = model.render(400, 400, t(K), 1024, rays=t(rays), ndc=True, near=0., far=1., use_viewdirs=True, N_samples=64,
                                       N_importance=32, retraw=True, perturb=0., raw_noise_std=0.)
    out.update(c_rgb=n(rgb), c_depth=n(depth), c_acc=n(acc), **{f"c_{k}": n(v) for k, v in ex.items()})
    save("G24_render_other_multires", **out)


def G25_pbe_composite_feature():
    """kernel_type='PBE': the coarse PDRF level composites its geo features before the colour network (renderer.py:30-34,
    voxnerf.py:223-239).  Eval render (64 + 32) and coarse_render (renderer.py:468-592: rgb + the composited feature map the PBE
    blur kernel consumes) on the small grids of G9."""
    K = W.synthetic_camera()
    model, _ = _nerfall("c2f", 32, 0, rgb_add_bias=False, kernel_type="PBE", **PDRF_SMALL)
    assert model.mlp_coarse.composite_feature and not model.mlp_fine.composite_feature
    ref_import.load_np_state_dict(model, _pdrf_sds(model, 91, 92))
    model.train(False)
    rays = W.synthetic_rays(51, 56)
    rgb, depth, acc, ex = model.render(400, 400, t(K), 1024, rays=t(rays), ndc=True, near=0., far=1., use_viewdirs=True, N_samples=64,
                                       N_importance=32, retraw=True, perturb=0., raw_noise_std=0.)
    out = dict(rgb=n(rgb), depth=n(depth), acc=n(acc), **{k: n(v) for k, v in ex.items()})
    crgb, cfeat = model.coarse_render(400, 400, t(K), 1024, rays=t(rays), ndc=True, near=0., far=1., use_viewdirs=True, N_samples=64,
                                      N_importance=32, perturb=0., raw_noise_std=0.)
    out.update(coarse_rgb=n(crgb), coarse_feat=n(cfeat))
    # the level alone on explicit inputs (VoxelNeRFBase.forward, both outputs)
    rs = np.random.RandomState(909)
    R, S = 24, 40
    pts = (rs.uniform(-1.0, 1.0, size=(R, S, 3)) * np.array([1.4, 1.4, 0.9])).astype(np.float32)
    z = np.sort(rs.uniform(0.0, 1.0, size=(R, S)).astype(np.float32), -1)
    vd = rs.normal(size=(R, 3)).astype(np.float32)
    vd /= np.linalg.norm(vd, axis=-1, keepdims=True)
    rd = (vd * rs.uniform(0.8, 1.3, size=(R, 1))).astype(np.float32)
    fts = model.mlp_coarse.sample(t(pts))
    col, dep, ac, wts, fm = model.mlp_coarse(t(pts), t(vd), fts, model.embed_fn, model.embeddirs_fn, t(z), t(rd), 0., False)
    out.update(l_pts=pts, l_z=z, l_vd=vd, l_rd=rd, l_fts=n(fts), l_color=n(col), l_depth=n(dep), l_acc=n(ac), l_weights=n(wts), l_feature=n(fm))
    # mode='nerf' with kernel_type PBE: the coarse network's feature map is composited (nerf.py:167-169)
    modeln, _ = _nerfall("nerf", 0, 0, kernel_type="PBE")
    assert modeln.mlp_coarse.composite_feature
    ref_import.load_np_state_dict(modeln, W.prefixed(W.make_nerf_state_dict(93), "mlp_coarse"))
    modeln.train(False)
    nrgb, nfeat = modeln.coarse_render(400, 400, t(K), 1024, rays=t(rays[:24]), ndc=True, near=0., far=1., use_viewdirs=True, N_samples=48,
                                       perturb=0., raw_noise_std=0.)
    out.update(nerf_coarse_rgb=n(nrgb), nerf_coarse_feat=n(nfeat))
    save("G25_pbe_composite_feature", **out)


def G8_appfeature():
    model, _ = _nerfall("c2f", 64, 0, rgb_add_bias=False, **PDRF_SMALL)
    sd = _pdrf_sds(model, 21, 22)
    ref_import.load_np_state_dict(model, sd)
    gc = [int(v) for v in model.mlp_coarse.gridSize]
    gf = [int(v) for v in model.mlp_fine.gridSize]
    assert gc == W.pdrf_grid_size([-1.5, -1.5, -1.0], [1.5, 1.5, 1.0], 24 ** 3), gc
    assert gf == W.pdrf_grid_size([-1.5, -1.5, -1.0], [1.5, 1.5, 1.0], 48 ** 3), gf
    rs = np.random.RandomState(808)
    pts = rs.uniform(-1.0, 1.0, size=(6, 40, 3)).astype(np.float32) * np.array([1.5, 1.5, 1.0], np.float32)
    pts[0, :8] *= 1.3                      # outside the box: zero padding
    pts[1, 0] = [-1.5, -1.5, -1.0]         # exactly on the corners
    pts[1, 1] = [1.5, 1.5, 1.0]
    pts[1, 2] = [0.0, 0.0, 0.0]
    fc = model.mlp_coarse.sample(t(pts))
    ff = model.mlp_fine.sample(t(pts))
    save("G8_appfeature", pts=pts, grid_coarse=np.array(gc), grid_fine=np.array(gf), ft_coarse=n(fc), ft_fine=n(ff))


def G9_render_c2f():
    K = W.synthetic_camera()
    model, _ = _nerfall("c2f", 64, 0, rgb_add_bias=False, **PDRF_SMALL)
    ref_import.load_np_state_dict(model, _pdrf_sds(model, 31, 32))
    model.train(False)
    rays = W.synthetic_rays(9, 64)
    rgb, depth, acc, ex = model.render(400, 400, t(K), 1024, rays=t(rays), ndc=True, near=0., far=1.,
                                       use_viewdirs=True, N_samples=64, N_importance=64, retraw=True,
                                       perturb=0., raw_noise_std=0.)
    out = dict(rgb=n(rgb), depth=n(depth), acc=n(acc), **{k: n(v) for k, v in ex.items()})
    # coarse only
    model0, _ = _nerfall("c2f", 0, 0, rgb_add_bias=False, **PDRF_SMALL)
    ref_import.load_np_state_dict(model0, _pdrf_sds(model0, 31, 32))
    model0.train(False)
    rgb, depth, acc, ex = model0.render(400, 400, t(K), 1024, rays=t(rays), ndc=True, near=0., far=1.,
                                        use_viewdirs=True, N_samples=64, N_importance=0, retraw=True,
                                        perturb=0., raw_noise_std=0.)
    out.update(c_rgb=n(rgb), c_depth=n(depth), c_acc=n(acc), c_weights=n(ex["weights"]))
    # per-sample features of the fine level (what AWP consumes): first 8 of 128 channels
    model.use_awp = True
    rgb, depth, acc, ex = model.render(400, 400, t(K), 1024, rays=t(rays[:16]), ndc=True, near=0., far=1.,
                                       use_viewdirs=True, N_samples=64, N_importance=64, retraw=True,
                                       perturb=0., raw_noise_std=0.)
    out.update(f_depth_feature=n(ex["depth_feature"])[:, :, :8], f_rays_d=n(ex["rays_d"]))
    save("G9_render_c2f", **out)


def G10_rbk_weighted_sum():
    from networks.dpnerf.blurmodel import RigidBlurringModel
    rs = np.random.RandomState(1010)
    R, P, S = 24, 10, 16
    rgb = rs.uniform(0, 1, size=(R * P, 3)).astype(np.float32)
    depth = rs.uniform(0, 1, size=(R * P,)).astype(np.float32)
    acc = rs.uniform(0, 1, size=(R * P,)).astype(np.float32)
    ex = {"rgb0": rs.uniform(0, 1, size=(R * P, 3)).astype(np.float32),
          "z_std": rs.uniform(0, 1, size=(R * P,)).astype(np.float32),
          "weights": rs.uniform(0, 1, size=(R * P, S)).astype(np.float32),
          "depth_feature": rs.standard_normal((R * P, S, 4)).astype(np.float32)}
    logits = rs.standard_normal((R, P)).astype(np.float32)
    ccw = 1.0 / (1.0 + np.exp(-logits))
    ccw = (ccw / ccw.sum(-1, keepdims=True)).astype(np.float32)
    fake_self = type("S", (), {"num_motion": P - 1, "use_origin": True})()
    o_rgb, o_depth, o_acc, o_ex = RigidBlurringModel.rbk_weighted_sum(
        fake_self, t(rgb), t(depth), t(acc), {k: t(v) for k, v in ex.items()}, t(ccw))
    save("G10_rbk_weighted_sum", rgb=rgb, depth=depth, acc=acc, ccw=ccw, **{f"ex_{k}": v for k, v in ex.items()},
         o_rgb=n(o_rgb), o_depth=n(o_depth), o_acc=n(o_acc), **{f"o_{k}": n(v) for k, v in o_ex.items()})


def _tonemap(map_rgb, map_ev, extra_ev, seed):
    from networks.tonemapping import TonemappingTransform
    tm = TonemappingTransform(map_rgb, map_ev, init_learn_identity=False, extra_features_event=extra_ev)
    if map_ev == "learn":
        ref_import.load_np_state_dict(tm.tonemapping_event, W.make_crf_state_dict(seed, extra_ev))
    if map_rgb == "learn":
        ref_import.load_np_state_dict(tm.tonemapping_rgb, W.make_crf_state_dict(seed + 1, 0))
    return tm


def G11_crf():
    rs = np.random.RandomState(1111)
    x = rs.uniform(0, 1, size=(200, 3)).astype(np.float32)
    x[0] = 0.0
    x[1] = 1.0
    f2 = np.stack([-rs.randint(0, 4, 200), rs.randint(0, 4, 200)], -1).astype(np.float32)
    f32 = np.stack([-rs.randint(0, 4, (200, 3)), rs.randint(0, 4, (200, 3))], -1).astype(np.float32)
    out = dict(x=x, f2=f2, f32=f32)
    tm = _tonemap("gamma", "learn", 2, 41)
    out["rgb_gamma"] = n(tm(t(x), mode="encode_rgb"))
    out["luma_learn_f2"] = n(tm(t(x), mode="encode_luma", ev_extra_feat=t(f2)))
    out["luma_learn_nofeat"] = n(tm(t(x), mode="encode_luma"))
    out["luma_learn_skip"] = n(tm(t(x), mode="encode_luma", skip_learn_crf=True, ev_extra_feat=t(f2)))
    out["tone_learn_f32"] = n(tm(t(x), mode="encode_luma", tonemap_only=True, ev_extra_feat=t(f32)))
    out["luma_learn_keep"] = n(tm(t(x), mode="encode_luma", keep_rgb=True, ev_extra_feat=t(f2)))
    out["luma_chunked"] = n(tm(t(x), mode="encode_luma", chunk=64, ev_extra_feat=None))
    tm = _tonemap("none", "learn", 0, 43)
    out["rgb_none"] = n(tm(t(x), mode="encode_rgb"))
    out["luma_learn0"] = n(tm(t(x), mode="encode_luma"))
    tm = _tonemap("gamma", "gamma", 0, 45)
    out["luma_gamma"] = n(tm(t(x), mode="encode_luma"))
    for std in ("rec709", "avg"):
        from networks.tonemapping import TonemappingTransform
        tm = TonemappingTransform("gamma", "gamma", luma_standard=std)
        out[f"luma_gamma_{std}"] = n(tm(t(x), mode="encode_luma"))
    save("G11_crf", **out)


def G12_egm_loss():
    from utils.events import egm_loss
    rs = np.random.RandomState(1212)
    N = 300
    ls = rs.uniform(0.01, 1, size=(N, 1)).astype(np.float32)
    le = rs.uniform(0.01, 1, size=(N, 1)).astype(np.float32)
    ls3 = rs.uniform(0.0, 1, size=(N, 3)).astype(np.float32)
    le3 = rs.uniform(0.0, 1, size=(N, 3)).astype(np.float32)
    bii = (0.2 * rs.randint(-3, 4, N)).astype(np.float32)
    cidx = rs.randint(0, 3, N)
    cmask = np.zeros((N, 3), dtype=bool)
    cmask[np.arange(N), cidx] = True
    out = dict(ls=ls, le=le, ls3=ls3, le3=le3, bii=bii, cmask=cmask)
    out["loss_plain"] = n(egm_loss(t(ls), t(le), t(bii)))
    out["loss_mask"] = n(egm_loss(t(ls3), t(le3), t(bii), color_mask=t(cmask)))
    out["loss_mask_w"] = n(egm_loss(t(ls3), t(le3), t(bii), color_mask=t(cmask), color_weight=[0.4, 0.2, 0.4]))
    save("G12_egm_loss", **out)


def G13_edi():
    from utils import edi
    rs = np.random.RandomState(1313)
    H = Wd = 16
    steps = 9
    bii = []
    ev = {}
    for s in range(steps - 1):
        ne = rs.randint(20, 60)
        x = rs.uniform(0, Wd - 1, ne).astype(np.float32)
        y = rs.uniform(0, H - 1, ne).astype(np.float32)
        x[:3] = np.floor(x[:3])              # integer coordinates: floor/ceil de-dup rule
        y[1:4] = np.floor(y[1:4])
        x[4] = Wd - 1                        # last column: ceil falls outside
        p = rs.randint(0, 2, ne).astype(np.int8)
        img = edi.brightness_increment_image(x, y, p, Wd, H, 0.2, 0.25, interpolate=True)
        img_ni = edi.brightness_increment_image(np.floor(x), np.floor(y), p, Wd, H, 0.2, 0.25, interpolate=False)
        ev[f"x{s}"], ev[f"y{s}"], ev[f"p{s}"] = x, y, p
        ev[f"bii_ni{s}"] = img_ni
        bii.append(img)
    bii = np.stack(bii)
    blurry = rs.uniform(0.05, 1, size=(H, Wd)).astype(np.float32)
    inner = edi.inner_double_integral(bii)
    sharp = edi.deblur_double_integral(blurry, bii)
    bii3 = np.stack([bii, 0.5 * bii, -bii], -1)
    blurry3 = rs.uniform(0.05, 1, size=(H, Wd, 3)).astype(np.float32)
    sharp3 = edi.deblur_double_integral(blurry3, bii3)
    slow = np.stack(edi.slowmo_double_integral(sharp, bii))
    save("G13_edi", bii=bii, blurry=blurry, inner=inner, sharp=sharp, bii3=bii3, blurry3=blurry3, sharp3=sharp3,
         slowmo=slow, **ev)


def G14_loss_assembly():
    """One synthetic step of the loss block (reference run_nerf.py:443-497, 518-591) with the
    renderer outputs replaced by given tensors; restates only the *composition order*, every
    operator is the reference's own (TonemappingTransform, egm_loss, img2mse)."""
    from utils.events import egm_loss
    sys.modules.setdefault("skimage", type(sys)("skimage"))
    sys.modules["skimage"].metrics = None
    sys.modules.setdefault("networks.lpips", type(sys)("networks.lpips"))
    sys.modules["networks.lpips"].LPIPS = None
    from utils.metrics import img2mse
    rs = np.random.RandomState(1414)
    out = {}
    for cfg in ("blender", "cdavis"):
        R, P, NE = 64, 10, (96 if cfg == "blender" else 80)
        rgb_p = rs.uniform(0, 1, size=(R * P, 3)).astype(np.float32)     # fine rgb of every sub-exposure ray
        rgb0_p = rs.uniform(0, 1, size=(R * P, 3)).astype(np.float32)    # coarse
        logits = rs.standard_normal((2, R, P)).astype(np.float32)
        ccw = 1.0 / (1.0 + np.exp(-logits))
        ccw = (ccw / ccw.sum(-1, keepdims=True)).astype(np.float32)      # [0]=weight1, [1]=ccw_fine (AWP)
        target = rs.uniform(0, 1, size=(R, 3)).astype(np.float32)
        target_pts0 = rs.uniform(0, 1, size=(R, 3)).astype(np.float32)
        fine_loss_weight, w_pts0, w_egm, w_tv, tv = 0.1, 0.01, 0.1, 1.0, 0.0
        tm = _tonemap("gamma" if cfg == "blender" else "none", "learn", 2, 51)
        crf = lambda x, **k: tm(x, **k)
        rgb = (t(rgb_p).reshape(R, P, 3) * t(ccw[0])[..., None]).sum(1)
        rgb1 = (t(rgb0_p).reshape(R, P, 3) * t(ccw[0])[..., None]).sum(1)
        rgb_awp = (t(rgb_p).reshape(R, P, 3) * t(ccw[1])[..., None]).sum(1)
        loss = img2mse(crf(rgb, mode="encode_rgb"), t(target)) + img2mse(crf(rgb1, mode="encode_rgb"), t(target))
        img_fine = img2mse(crf(rgb_awp, mode="encode_rgb"), t(target))
        loss = loss * (1 - fine_loss_weight) + img_fine * fine_loss_weight
        pts0 = 0.0
        for x in (t(rgb_p).reshape(R, P, 3)[:, 0], t(rgb0_p).reshape(R, P, 3)[:, 0]):
            pts0 = pts0 + img2mse(crf(x, mode="encode_rgb"), t(target_pts0))
        loss = loss + pts0 * w_pts0
        # events
        thr = np.array([0.2, 0.2] if cfg == "blender" else [0.25, 0.25], np.float32)
        cn = -rs.randint(0, 4, NE).astype(np.float32)
        cp = rs.randint(0, 4, NE).astype(np.float32)
        bii = (t(thr) * torch.stack([t(cn), t(cp)], -1)).sum(-1)
        es, es0, ee, ee0 = (rs.uniform(0.02, 1, size=(NE, 3)).astype(np.float32) for _ in range(4))
        cidx = rs.randint(0, 3, NE)
        cmask = np.zeros((NE, 3), dtype=bool)
        cmask[np.arange(NE), cidx] = True
        if cfg == "blender":
            feat = torch.stack([t(cn), t(cp)], -1)
            kw, cm, cw = {}, None, None
        else:
            fn = torch.zeros(NE, 3)
            fp = torch.zeros(NE, 3)
            fn[t(cmask)] = t(cn)
            fp[t(cmask)] = t(cp)
            feat = torch.stack([fn, fp], -1)
            kw, cm, cw = {"tonemap_only": True}, t(cmask), [0.4, 0.2, 0.4]
        l_s = crf(t(es), mode="encode_luma", ev_extra_feat=feat, **kw)
        l_s0 = crf(t(es0), mode="encode_luma", ev_extra_feat=feat, **kw)
        l_e = crf(t(ee), mode="encode_luma", ev_extra_feat=feat, **kw)
        l_e0 = crf(t(ee0), mode="encode_luma", ev_extra_feat=feat, **kw)
        egm = egm_loss(l_s0, l_e0, bii, color_mask=cm, color_weight=cw) + \
            egm_loss(l_s, l_e, bii, color_mask=cm, color_weight=cw)
        total = loss + tv * w_tv + egm * w_egm
        out.update({f"{cfg}_rgb_p": rgb_p, f"{cfg}_rgb0_p": rgb0_p, f"{cfg}_ccw": ccw, f"{cfg}_target": target,
                    f"{cfg}_target_pts0": target_pts0, f"{cfg}_cn": cn, f"{cfg}_cp": cp, f"{cfg}_cmask": cmask,
                    f"{cfg}_es": es, f"{cfg}_es0": es0, f"{cfg}_ee": ee, f"{cfg}_ee0": ee0,
                    f"{cfg}_img_loss": n(loss), f"{cfg}_pts0": n(pts0), f"{cfg}_egm": n(egm), f"{cfg}_total": n(total),
                    f"{cfg}_scalars": np.array([fine_loss_weight, w_pts0, w_egm], np.float32)})
    save("G14_loss_assembly", **out)


def G15_awp_feature_integration():
    """AdaptiveWeightProposal.feature_integration (networks/dpnerf/awp.py:49-77): per-channel compositing scan of the
    per-sample embedded features (each channel is its own density; a ZERO alpha is appended, not a one)."""
    from networks.dpnerf.awp import AdaptiveWeightProposal
    awp = AdaptiveWeightProposal(input_ch=128, num_motion=9, D_sam=4, W_sam=64, D_mot=1, W_mot=32, dir_freq=2, rgb_freq=2,
                                 depth_freq=3, ray_dir_freq=2, view_feature_ch=32)
    rs = np.random.RandomState(151)
    out = {}
    for tag, (R, P, S, Cc) in {"a": (2, 3, 128, 64), "b": (3, 2, 33, 16)}.items():
        feat = np.abs(rs.standard_normal((R, P, S, Cc))).astype(np.float32) * rs.choice([0.0, 0.5, 4.0, 60.0], size=(R, P, S, 1)).astype(np.float32)
        z = np.sort(rs.uniform(0, 1, (R * P, S)).astype(np.float32), -1)
        z[:, 5] = z[:, 4]                                   # a duplicate z (zero interval)
        rays_d = rs.standard_normal((R * P, 3)).astype(np.float32)
        res = awp.feature_integration(t(feat), t(z), t(rays_d))
        out.update({f"{tag}_feat": feat, f"{tag}_z": z, f"{tag}_rays_d": rays_d, f"{tag}_out": n(res)})
    save("G15_awp_feature_integration", **out)


def G16_rbk_warp():
    """RigidBlurringModel.rbk_warp (networks/dpnerf/blurmodel.py:51-82 with SE3Field / RigidBody of utils/rigid_warping.py):
    rays [R,3,2] + predicted screw motions r, v [R, 3*M] -> warped rays [R, M(+1), 3, 2] and the 4x4 transforms."""
    from networks.dpnerf.blurmodel import RigidBlurringModel
    rs = np.random.RandomState(161)
    out = {}
    for tag, (R, M, uo, scale) in {"a": (64, 9, True, 1e-2), "b": (16, 4, False, 1.0), "c": (8, 9, True, 1e-5)}.items():
        net = RigidBlurringModel(W=32, D_r=1, W_r=16, D_v=1, W_v=16, D_w=1, W_w=16, output_ch_r=3, output_ch_v=3, feat_ch=0,
                                 rv_window=1.0, view_embed=None, num_motion=M, use_origin=uo, use_view_embed=False)
        rays = W.synthetic_rays(170 + len(out), R)
        r = (rs.standard_normal((R, 3 * M)) * scale).astype(np.float32)
        v = (rs.standard_normal((R, 3 * M)) * scale).astype(np.float32)
        if tag == "a":
            r[0] = 0.0                                     # theta = 1e-10: the pure-translation limit
        new_rays, tf = net.rbk_warp(t(rays), t(r), t(v), return_transform=True)
        out.update({f"{tag}_rays": rays, f"{tag}_r": r, f"{tag}_v": v, f"{tag}_new_rays": n(new_rays), f"{tag}_transform": n(tf)})
    save("G16_rbk_warp", **out)


def G17_compute_successor():
    """utils/events.py:72-120 compute_successor (numba @njit in the reference; run here as plain Python through the njit stub):
    per event the index of the next event at the same pixel, the number of later events there, and per pixel the first /
    last event index.  Integer outputs: the parity bar is bit-exact."""
    from utils.events import compute_successor
    rs = np.random.RandomState(171)
    out = {}
    for tag, (N, hw, flat) in {"a": (6000, 37 * 23, True), "b": (500, 9, True), "c": (1, 4, True)}.items():
        ids = rs.randint(0, hw, size=N).astype(np.int64)
        if tag == "a":
            ids[:200] = 5                                    # a long chain at one pixel
        ids[-1] = hw - 1                                     # the reference sizes its tables by max(id) + 1
        ev = np.stack([ids, np.sort(rs.uniform(0, 1, N)), rs.choice([-1, 1], N)], -1).astype(np.float64)
        succ, nsucc, latest, first = compute_successor(ev, flat_xy=flat)
        out.update({f"{tag}_ids": ids.astype(np.int32), f"{tag}_succ": succ.astype(np.int64), f"{tag}_nsucc": nsucc.astype(np.int32),
                    f"{tag}_latest": latest.reshape(-1).astype(np.int64), f"{tag}_first": first.reshape(-1).astype(np.int64)})
    save("G17_compute_successor", **out)


def _grad_summaries(named_grads, out, prefix, elements=True):
    """store (norm, seeded projection) + the first 32 elements of every gradient: full gradients would not fit a small fixture"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from torch_restatement import grad_elements, grad_summary
    for i, (name, g) in enumerate(named_grads):
        sm, head = grad_summary(n(g), 7000 + i)
        out[f"{prefix}{name}.summary"] = sm.astype(np.float64)
        out[f"{prefix}{name}.head"] = head.astype(np.float64)
        if not elements:
            continue
        idx, val = grad_elements(n(g), 9000 + i)                # element-level pins: the 256 largest elements + 256 seeded random ones
        out[f"{prefix}{name}.elem_idx"] = idx
        out[f"{prefix}{name}.elem_val"] = val.astype(np.float64)


def G18_nerf_grads():
    """Gradients of the reference NeRF (D 8, W 64 to keep the fixture small; same code path as W 256) through mlpforward +
    raw2outputs, w.r.t. every parameter and the inputs: torch.autograd on the reference modules themselves."""
    from networks.embedding import get_embedder
    rs = np.random.RandomState(1801)
    R, S = 12, 10
    o = rs.uniform(-0.4, 0.4, size=(R, 3)).astype(np.float32)
    d = rs.standard_normal((R, 3)).astype(np.float32)
    z = np.sort(rs.uniform(0.2, 1.8, size=(R, S)).astype(np.float32), -1)
    w_rgb = rs.standard_normal((R, 3)).astype(np.float32)
    net = _ref_nerf(18, D=8, Wd=64, rgb_add_bias=True)
    net.train(True)
    e10, _ = get_embedder(10)
    e4, _ = get_embedder(4)
    with torch.enable_grad():
        ot, dt = t(o).requires_grad_(True), t(d).requires_grad_(True)
        pts = ot[:, None] + dt[:, None] * t(z)[..., None]
        vd = dt / torch.norm(dt, dim=-1, keepdim=True)
        x = torch.cat([e10(pts.reshape(-1, 3)), e4(vd[:, None].expand(-1, S, -1).reshape(-1, 3))], -1)
        raw, _ = net.eval(x)
        rgb_map = net.raw2outputs(raw.reshape(R, S, 4), t(z), dt)[0]
        loss = (rgb_map * t(w_rgb)).sum()
        params = list(net.named_parameters())
        grads = torch.autograd.grad(loss, [p for _, p in params] + [ot, dt])
    out = {}
    _grad_summaries([(k, g) for (k, _), g in zip(params, grads)] + [("rays_o", grads[-2]), ("rays_d", grads[-1])], out, "g.")
    # the same on the full-width network (what the HIP training path is built for): summaries only
    net2 = _ref_nerf(19, D=8, Wd=256, rgb_add_bias=True)
    net2.train(True)
    with torch.enable_grad():
        ot, dt = t(o).requires_grad_(True), t(d).requires_grad_(True)
        pts = ot[:, None] + dt[:, None] * t(z)[..., None]
        vd = dt / torch.norm(dt, dim=-1, keepdim=True)
        x = torch.cat([e10(pts.reshape(-1, 3)), e4(vd[:, None].expand(-1, S, -1).reshape(-1, 3))], -1)
        raw2, _ = net2.eval(x)
        rgb_map2 = net2.raw2outputs(raw2.reshape(R, S, 4), t(z), dt)[0]
        loss2 = (rgb_map2 * t(w_rgb)).sum()
        params2 = list(net2.named_parameters())
        grads2 = torch.autograd.grad(loss2, [p for _, p in params2] + [ot, dt])
    _grad_summaries([(k, g) for (k, _), g in zip(params2, grads2)] + [("rays_o", grads2[-2]), ("rays_d", grads2[-1])], out, "w256.")
    save("G18_nerf_grads", o=o, d=d, z=z, w_rgb=w_rgb, rgb_map=n(rgb_map), loss=np.float64(loss.item()), rgb_map_w256=n(rgb_map2), **out)


def _c2f_grads(name, R, ray_seed, rs_seed):
    K = W.synthetic_camera()
    model, _ = _nerfall("c2f", 16, 0, rgb_add_bias=True, **PDRF_SMALL)
    gc = [int(v) for v in model.mlp_coarse.gridSize]
    gf = [int(v) for v in model.mlp_fine.gridSize]
    sd = W.prefixed(W.make_pdrf_state_dict(91, gc, input_ch=32 + 63, hidden_dim=64, geo_feat_dim=15, add_bias_color=True), "mlp_coarse")
    sd.update(W.prefixed(W.make_pdrf_state_dict(92, gf, input_ch=64 + 63, hidden_dim=256, geo_feat_dim=128, add_bias_color=True), "mlp_fine"))
    ref_import.load_np_state_dict(model, sd)
    model.train(True)
    rays = W.synthetic_rays(ray_seed, R)
    rs = np.random.RandomState(rs_seed)
    w_rgb, w_rgb0 = rs.standard_normal((R, 3)).astype(np.float32), rs.standard_normal((R, 3)).astype(np.float32)
    with torch.enable_grad():
        rt = t(rays).requires_grad_(True)
        rgb, depth, acc, ex = model.render(400, 400, t(K), 1 << 20, rays=rt, ndc=True, near=0., far=1., use_viewdirs=True, N_samples=16,
                                           N_importance=16, retraw=True, perturb=0., raw_noise_std=0.)
        tv = (model.mlp_coarse.TV_loss_app() + model.mlp_fine.TV_loss_app()) * 5
        loss = (rgb * t(w_rgb)).sum() + (ex["rgb0"] * t(w_rgb0)).sum() + 0.1 * tv
        params = [(k, p) for k, p in model.named_parameters() if p.requires_grad]
        grads = torch.autograd.grad(loss, [p for _, p in params] + [rt], allow_unused=True)
    out = {}
    named = [(k, g) for (k, _), g in zip(params, grads) if g is not None] + [("rays", grads[-1])]
    _grad_summaries(named, out, "g.")
    extra = dict(z_vals=n(ex["z_vals"]), z_vals0=n(ex["z_vals0"])) if R <= 64 else {}
    save(name, rays=rays, w_rgb=w_rgb, w_rgb0=w_rgb0, rgb=n(rgb), rgb0=n(ex["rgb0"]), **extra,
         tv=np.float64(tv.item()), loss=np.float64(loss.item()), grid_coarse=np.array(gc), grid_fine=np.array(gf), **out)


def G19_c2f_grads():
    """Gradients of the reference's mode='c2f' training forward (NeRFAll.render in train mode: NDC ray packing, both PDRF levels,
    hierarchical resampling, TV regulariser) w.r.t. every parameter of both levels and the rays."""
    _c2f_grads("G19_c2f_grads", 24, 19, 1901)


def G30_c2f_grads_16k():
    """The G19 measurement at a batch where the ReLU-flip floor of a reduced-precision forward is visible as a statistic, not as one
    unlucky unit: 512 rays x (16 + 16) = 16 384 fine-level samples through the reference's whole mode='c2f' training forward under
    torch.autograd.  Same weights and grids as G19; summaries + 512 element pins per tensor."""
    _c2f_grads("G30_c2f_grads_16k", 512, 30, 3001)


def G20_loss_grads():
    """Gradients of the loss block of G14 (the reference's TonemappingTransform / CRF, egm_loss, img2mse under torch.autograd; only the
    composition order of run_nerf.py:443-497,518-591 is restated) w.r.t. the rendered colours, both weight sets and every parameter of
    the learnable event-CRF."""
    from utils.events import egm_loss
    sys.modules.setdefault("skimage", type(sys)("skimage"))
    sys.modules["skimage"].metrics = None
    sys.modules.setdefault("networks.lpips", type(sys)("networks.lpips"))
    sys.modules["networks.lpips"].LPIPS = None
    from utils.metrics import img2mse
    rs = np.random.RandomState(2020)
    out = {}
    for cfg in ("blender", "cdavis"):
        R, P, NE = 48, 10, (70 if cfg == "blender" else 61)
        rgb_p = rs.uniform(0.02, 1, size=(R * P, 3)).astype(np.float32)
        rgb0_p = rs.uniform(0.02, 1, size=(R * P, 3)).astype(np.float32)
        logits = rs.standard_normal((2, R, P)).astype(np.float32)
        ccw = 1.0 / (1.0 + np.exp(-logits))
        ccw = (ccw / ccw.sum(-1, keepdims=True)).astype(np.float32)
        target = rs.uniform(0, 1, size=(R, 3)).astype(np.float32)
        target_pts0 = rs.uniform(0, 1, size=(R, 3)).astype(np.float32)
        fine_loss_weight, w_pts0, w_egm = 0.1, 0.01, 0.1
        tm = _tonemap("gamma" if cfg == "blender" else "none", "learn", 2, 51)
        for prm in tm.tonemapping_event.parameters():          # away from the near-identity initialisation: real gradients in every layer
            if prm.dim() == 2:
                prm.data.mul_(3.0)
        thr = np.array([0.2, 0.2] if cfg == "blender" else [0.25, 0.25], np.float32)
        cn = -rs.randint(0, 4, NE).astype(np.float32)
        cp = rs.randint(0, 4, NE).astype(np.float32)
        es, es0, ee, ee0 = (rs.uniform(0.05, 0.95, size=(NE, 3)).astype(np.float32) for _ in range(4))
        cidx = rs.randint(0, 3, NE)
        cmask = np.zeros((NE, 3), dtype=bool)
        cmask[np.arange(NE), cidx] = True
        with torch.enable_grad():
            leaves = {k: t(v).requires_grad_(True) for k, v in dict(rgb_p=rgb_p, rgb0_p=rgb0_p, w1=ccw[0], w2=ccw[1], es=es, es0=es0, ee=ee, ee0=ee0).items()}
            crf = lambda x, **k: tm(x, **k)
            rp, r0p = leaves["rgb_p"].reshape(R, P, 3), leaves["rgb0_p"].reshape(R, P, 3)
            rgb = (rp * leaves["w1"][..., None]).sum(1)
            rgb1 = (r0p * leaves["w1"][..., None]).sum(1)
            rgb_awp = (rp * leaves["w2"][..., None]).sum(1)
            loss = img2mse(crf(rgb, mode="encode_rgb"), t(target)) + img2mse(crf(rgb1, mode="encode_rgb"), t(target))
            img_fine = img2mse(crf(rgb_awp, mode="encode_rgb"), t(target))
            loss = loss * (1 - fine_loss_weight) + img_fine * fine_loss_weight
            pts0 = 0.0
            for x in (rp[:, 0], r0p[:, 0]):
                pts0 = pts0 + img2mse(crf(x, mode="encode_rgb"), t(target_pts0))
            loss = loss + pts0 * w_pts0
            bii = (t(thr) * torch.stack([t(cn), t(cp)], -1)).sum(-1)
            if cfg == "blender":
                feat = torch.stack([t(cn), t(cp)], -1)
                kw, cm, cw = {}, None, None
            else:
                fn = torch.zeros(NE, 3)
                fp = torch.zeros(NE, 3)
                fn[t(cmask)] = t(cn)
                fp[t(cmask)] = t(cp)
                feat = torch.stack([fn, fp], -1)
                kw, cm, cw = {"tonemap_only": True}, t(cmask), [0.4, 0.2, 0.4]
            lum = lambda x: crf(x, mode="encode_luma", ev_extra_feat=feat, **kw)
            egm = egm_loss(lum(leaves["es0"]), lum(leaves["ee0"]), bii, color_mask=cm, color_weight=cw) + \
                egm_loss(lum(leaves["es"]), lum(leaves["ee"]), bii, color_mask=cm, color_weight=cw)
            total = loss + egm * w_egm
            crf_params = list(tm.tonemapping_event.named_parameters())
            grads = torch.autograd.grad(total, list(leaves.values()) + [p for _, p in crf_params])
        for (k, _), gr in zip(list(leaves.items()) + [("crf." + kk, vv) for kk, vv in crf_params], grads):
            out[f"{cfg}_g.{k}"] = n(gr)
        out.update({f"{cfg}_rgb_p": rgb_p, f"{cfg}_rgb0_p": rgb0_p, f"{cfg}_ccw": ccw, f"{cfg}_target": target, f"{cfg}_target_pts0": target_pts0,
                    f"{cfg}_cn": cn, f"{cfg}_cp": cp, f"{cfg}_cmask": cmask, f"{cfg}_es": es, f"{cfg}_es0": es0, f"{cfg}_ee": ee, f"{cfg}_ee0": ee0,
                    f"{cfg}_total": n(total), f"{cfg}_scalars": np.array([fine_loss_weight, w_pts0, w_egm], np.float32)})
    save("G20_loss_grads", **out)


def G21_awp_sample_embed():
    """AdaptiveWeightProposal.forward (networks/dpnerf/awp.py:79-117) up to the input of motion_feature_embed_layer: the per-sample
    embedding MLP (:98-100), feature_integration (:102) and the concatenation with the view embedding (:104-105), run as the
    reference runs them (forward hooks capture the intermediates of the real forward), plus torch.autograd gradients of a fixed
    projection of the integrated features w.r.t. depth_feature and the embedding's parameters (what run_nerf.py:593-601 computes)."""
    from networks.dpnerf.awp import AdaptiveWeightProposal
    torch.manual_seed(21)
    M = 2
    awp = AdaptiveWeightProposal(input_ch=128, num_motion=M, D_sam=4, W_sam=64, D_mot=1, W_mot=32, dir_freq=2, rgb_freq=2,
                                 depth_freq=3, ray_dir_freq=2, view_feature_ch=32)
    awp.load_state_dict({k: t(v) for k, v in W.make_awp_embed_state_dict(211).items()}, strict=False)
    rs = np.random.RandomState(212)
    R, P, S = 3, M + 1, 24
    depth_feature = (rs.standard_normal((R * P, S, 128)) * 0.7).astype(np.float32)
    z = np.sort(rs.uniform(0, 1, (R * P, S)).astype(np.float32), -1)
    rays_d = rs.standard_normal((R * P, 3)).astype(np.float32)
    view_feature = rs.standard_normal((R, 32)).astype(np.float32)
    proj = rs.standard_normal((R, P, 64)).astype(np.float32)
    cap = {}
    h1 = awp.sample_feature_embed_layer[-1].register_forward_hook(lambda m, i, o: cap.__setitem__("pre", o))
    h2 = awp.motion_feature_embed_layer[0].register_forward_pre_hook(lambda m, i: cap.__setitem__("h", i[0]))
    with torch.enable_grad():
        df = t(depth_feature).requires_grad_(True)
        out = awp(df, t(z), t(rays_d), t(view_feature))
        h = cap["h"]                                           # [R, P, 64 + 32 + ray_dirs_embed_ch]
        loss = (h[..., :64] * t(proj)).sum()
        params = [p for l in awp.sample_feature_embed_layer for p in (l.weight, l.bias)]
        grads = torch.autograd.grad(loss, [df] + params)
    h1.remove()
    h2.remove()
    res = {"depth_feature": depth_feature, "z": z, "rays_d": rays_d, "view_feature": view_feature, "proj": proj,
           "h_local": n(torch.relu(cap["pre"])), "h": n(h), "out": n(out), "loss": n(loss), "g.depth_feature": n(grads[0])}
    for l in range(4):
        res[f"g.w{l}"], res[f"g.b{l}"] = n(grads[1 + 2 * l]), n(grads[2 + 2 * l])
    save("G21_awp_sample_embed", **res)


def G22_mam():
    """MotionAggregationModule.forward (networks/dpnerf/mam.py:66-83, CorrelationModule.forward :27-53) in training mode (BatchNorm on
    the batch statistics, as under run_nerf.py's train loop): x_global [R, P, 32], x_local [R P, S, 64] -> [R, P, 32].  Forward
    pre-hooks on Corr.conva / Corr.convb capture curver_inter [R, 32, P] and curves_intra [R, 32, S] (the results of the per-sample
    part, mam.py:29-33); torch.autograd gradients of a fixed projection of the output w.r.t. both inputs and the per-sample part's
    parameters."""
    from networks.dpnerf.mam import MotionAggregationModule
    torch.manual_seed(22)
    M = 3
    mam = MotionAggregationModule(in_channels=32, k=3, num_motion=M)
    mam.train()
    rs = np.random.RandomState(221)
    R, P, S = 5, M + 1, 24
    x_global = rs.standard_normal((R, P, 32)).astype(np.float32)
    x_local = np.maximum(rs.standard_normal((R * P, S, 64)), 0).astype(np.float32)        # a ReLU output, like h_local
    proj = rs.standard_normal((R, P, 32)).astype(np.float32)
    cap = {}
    h1 = mam.Corr.conva.register_forward_pre_hook(lambda m, i: cap.__setitem__("inter", i[0]))
    h2 = mam.Corr.convb.register_forward_pre_hook(lambda m, i: cap.__setitem__("intra", i[0]))
    with torch.enable_grad():
        xg, xl = t(x_global).requires_grad_(True), t(x_local).requires_grad_(True)
        out = mam(xg, xl)
        loss = (out * t(proj)).sum()
        params = [mam.linear.weight, mam.linear.bias, mam.Corr.line_conv_att.weight]
        grads = torch.autograd.grad(loss, [xg, xl] + params)
    h1.remove()
    h2.remove()
    res = {"x_global": x_global, "x_local": x_local, "proj": proj, "out": n(out), "inter": n(cap["inter"]), "intra": n(cap["intra"]),
           "g.x_global": n(grads[0]), "g.x_local": n(grads[1]), "g.linear.weight": n(grads[2]), "g.linear.bias": n(grads[3]),
           "g.line_conv_att.weight": n(grads[4])}
    for k, v in mam.state_dict().items():
        res["sd." + k] = n(v)
    save("G22_mam", **res)


def G27_awp_per_ray():
    """AdaptiveWeightProposal.forward (networks/dpnerf/awp.py:79-117) in training mode with the REAL MotionAggregationModule
    (mam.py:56-83): the whole forward as the reference runs it, from depth_feature.  A forward pre-hook on MAM captures its two inputs
    (x_global = the motion embedding's output, x_local = h_local, the ReLU output of the per-sample embedding); torch.autograd gradients
    of a fixed projection of the proposal weights w.r.t. h_local, rays_d, view_feature and every parameter behind the per-sample
    embedding (motion_feature_embed_layer, MAM.*, w_linear).  BatchNorm (Corr.convd.1) runs on the batch statistics and updates its
    running estimates (stored after the step); `out_eval` is the same module in eval mode afterwards (running estimates)."""
    from networks.dpnerf.awp import AdaptiveWeightProposal
    torch.manual_seed(27)
    M, VF = 3, 5
    awp = AdaptiveWeightProposal(input_ch=128, num_motion=M, D_sam=4, W_sam=64, D_mot=1, W_mot=32, dir_freq=2, rgb_freq=2,
                                 depth_freq=3, ray_dir_freq=2, view_feature_ch=VF)
    awp.load_state_dict({k: t(v) for k, v in W.make_awp_embed_state_dict(271).items()}, strict=False)
    rs = np.random.RandomState(272)
    bn = awp.MAM.Corr.convd[1]
    with torch.no_grad():                                      # a BatchNorm that is not at its initial values
        bn.weight.copy_(t(rs.uniform(0.5, 1.5, 32).astype(np.float32)))
        bn.bias.copy_(t((rs.standard_normal(32) * 0.2).astype(np.float32)))
        bn.running_mean.copy_(t((rs.standard_normal(32) * 0.1).astype(np.float32)))
        bn.running_var.copy_(t(rs.uniform(0.5, 2.0, 32).astype(np.float32)))
        for conv, k in ((awp.MAM.Corr.conva, 8.0), (awp.MAM.Corr.convb, 8.0), (awp.MAM.Corr.convc, 8.0), (awp.MAM.Corr.line_conv_att, 30.0)):
            conv.weight.mul_(k)                                # attention logits of order 1 (at the initial scale both softmaxes are uniform to 1e-3)
    sd_before = {k: n(v).copy() for k, v in awp.state_dict().items() if not k.startswith("sample_feature_embed_layer")}
    awp.train()
    R, P, S = 7, M + 1, 24
    depth_feature = (rs.standard_normal((R * P, S, 128)) * 2.5).astype(np.float32)
    z = np.sort(rs.uniform(0, 1, (R * P, S)).astype(np.float32), -1)
    rays_d = rs.standard_normal((R * P, 3)).astype(np.float32)
    view_feature = rs.standard_normal((R, VF)).astype(np.float32)
    proj = rs.standard_normal((R, P)).astype(np.float32)
    cap = {}
    hk = awp.MAM.register_forward_pre_hook(lambda m, i: cap.update(x_global=i[0], h_local=i[1]))
    names = [k for k, _ in awp.named_parameters() if not k.startswith("sample_feature_embed_layer") and not k.startswith("MAM.conv.")]
    pd = dict(awp.named_parameters())
    with torch.enable_grad():
        rd, vf = t(rays_d).requires_grad_(True), t(view_feature).requires_grad_(True)
        out = awp(t(depth_feature), t(z), rd, vf)
        loss = (out * t(proj)).sum()
        grads = torch.autograd.grad(loss, [cap["h_local"], rd, vf] + [pd[k] for k in names], allow_unused=True)
    hk.remove()
    res = {"z": z, "rays_d": rays_d, "view_feature": view_feature, "proj": proj, "h_local": n(cap["h_local"]), "x_global": n(cap["x_global"]),
           "out": n(out), "g.h_local": n(grads[0]), "g.rays_d": n(grads[1]), "g.view_feature": n(grads[2])}
    for k, g in zip(names, grads[3:]):
        if g is not None:                                      # (Corr.line_conv_att, MAM.linear: reached through the per-sample part too)
            res["g." + k] = n(g)
    for k, v in sd_before.items():
        res["sd." + k] = v
    for k in ("running_mean", "running_var", "num_batches_tracked"):
        res["after." + k] = n(getattr(bn, k))
    awp.eval()
    res["out_eval"] = n(awp(t(depth_feature), t(z), t(rays_d), t(view_feature)))
    save("G27_awp_per_ray", **res)


def G26_sample_events():
    """EventsDataset.sample_events (data/loader_events.py:259-304).  The class itself cannot be imported (np.bool under numpy 2, h5py);
    its two branches are the reference's own functions composed as the method composes them: compute_successor for the table's last
    column (:241-247), plain indexing for the single-successor branch (:272-276), gather_successor (utils/events.py:221-257) for the
    multi-hop branch (:262-271), get_rays_pix (utils/rays.py:25-36) on the start / end poses with add_halfpix = integer_coords
    (:292-297).  The per-event poses stand in for interpolate_poses (scipy, CPU): random rigid poses, an INPUT of the fixture."""
    from utils.events import compute_successor, gather_successor
    from utils.rays import get_rays_pix
    rs = np.random.RandomState(2601)
    out = {}
    K = W.synthetic_camera()
    for tag, (N, ncoord, nq, color, halfpix) in {"a": (1500, 37 * 23, 256, False, True), "b": (400, 11, 97, True, False)}.items():
        ids = rs.randint(0, ncoord, size=N).astype(np.int64)
        ids[:60] = 3                                                   # a long chain at one coordinate
        ids[-1] = ncoord - 1
        tms = np.sort(rs.uniform(0, 1e6, N))
        pol = rs.choice([-1.0, 1.0], N)
        ev3 = np.stack([ids, tms, pol], -1).astype(np.float64)
        succ, nsucc, _, _ = compute_successor(ev3, flat_xy=True)
        events = np.concatenate([ev3, succ.reshape(-1, 1)], -1)        # loader_events.py:247
        coords = np.stack([rs.uniform(0, 399, ncoord), rs.uniform(0, 399, ncoord)], -1).astype(np.float32)
        if halfpix:
            coords = np.floor(coords)
        cmap = None
        if color:
            cmap = np.zeros((ncoord, 3), bool)
            cmap[np.arange(ncoord), rs.randint(0, 3, ncoord)] = True
        # one rigid pose per event (what interpolate_poses returns for the event's timestamp)
        ax = rs.standard_normal((N, 3)) * 0.2
        th = np.linalg.norm(ax, axis=-1, keepdims=True)
        kx = ax / th
        Km = np.zeros((N, 3, 3))
        Km[:, 0, 1], Km[:, 0, 2], Km[:, 1, 0], Km[:, 1, 2], Km[:, 2, 0], Km[:, 2, 1] = -kx[:, 2], kx[:, 1], kx[:, 2], -kx[:, 0], -kx[:, 1], kx[:, 0]
        Rm = np.eye(3)[None] + np.sin(th)[..., None] * Km + (1 - np.cos(th))[..., None] * (Km @ Km)
        poses = np.concatenate([Rm, rs.uniform(-0.3, 0.3, (N, 3, 1))], -1).astype(np.float32)
        with_succ = np.where(nsucc > 0)[0]
        q = with_succ[rs.randint(0, with_succ.shape[0], nq)].astype(np.int64)
        ev_t, poses_t, coords_t = torch.tensor(events), t(poses), t(coords)
        start = ev_t[q]
        # ---- branch 1 (:272-276)
        end = ev_t[start[:, -1].long()]
        pmask = end[:, -2] > 0
        pos1 = torch.where(pmask, end[:, -2], 0)
        neg1 = torch.where(~pmask, end[:, -2], 0)
        cid = start[:, 0].long()
        cc = coords_t[cid]

        def rays(idx):
            return torch.stack(get_rays_pix(cc, t(K), poses_t[idx][:, :3, :4], add_halfpix=halfpix), 1).permute(0, 2, 1)
        out.update({f"{tag}_events": events, f"{tag}_coords": coords, f"{tag}_poses": poses, f"{tag}_ids": q, f"{tag}_halfpix": np.array(int(halfpix)),
                    f"{tag}_pos1": n(pos1).astype(np.float32), f"{tag}_neg1": n(neg1).astype(np.float32), f"{tag}_cid": n(cid).astype(np.int64),
                    f"{tag}_rays_start": n(rays(torch.tensor(q))), f"{tag}_rays_end1": n(rays(start[:, -1].long()))})
        if cmap is not None:
            out[f"{tag}_cmap"] = cmap
            out[f"{tag}_cmap_out"] = cmap[n(cid)]
        # ---- branch 2 (:262-271): hops in [0, num_successors - 1] (the caller's random draw), plus a few beyond the chain's end
        hops = np.minimum(rs.randint(0, 6, nq), nsucc[q] - 1).astype(np.int64)
        si, negp, posp = gather_successor(torch.tensor(q), torch.tensor(hops), ev_t[:, -1].long(), ev_t[:, -2].int())
        out.update({f"{tag}_hops": hops, f"{tag}_succ2": n(si).astype(np.int64), f"{tag}_pos2": n(posp).astype(np.float32), f"{tag}_neg2": n(negp).astype(np.float32),
                    f"{tag}_rays_end2": n(rays(si))})
    save("G26_sample_events", **out)

def _stub_voxels():
    """data/loader.py imports utils/voxels.py, whose module-level tensor is created on 'cuda' (:7): not importable without a GPU.  The
    loader only takes get_bbox3d_for_llff from it (used in __init__, not in the methods called here)."""
    import types
    if "utils.voxels" not in sys.modules:
        m = types.ModuleType("utils.voxels")
        m.get_bbox3d_for_llff = None
        sys.modules["utils.voxels"] = m


def G28_image_batch():
    """LLFFDataset.__getitem__ (data/loader.py:325-356) called UNBOUND on an object that carries the attributes the method reads
    (device, n_imgs, h, w, poses, images, K, pts0_images): the reference's own method body runs, the constructor (disk I/O) does not.
    Also get_rays / get_rays_pix with add_halfpix=False (utils/rays.py:8-36), the case of rectified event coordinates."""
    import types
    _stub_voxels()
    from data.loader import LLFFDataset
    from utils.rays import get_rays, get_rays_pix
    rs = np.random.RandomState(2801)
    n_img, H, W = 4, 23, 31
    images = rs.uniform(0, 1, (n_img, H, W, 3)).astype(np.float32)
    pts0 = rs.uniform(0, 1, (n_img, H, W, 3)).astype(np.float32)
    ax = rs.standard_normal((n_img, 3)) * 0.2
    from scipy.spatial.transform import Rotation as Rot
    poses = np.concatenate([Rot.from_rotvec(ax).as_matrix(), rs.uniform(-0.3, 0.3, (n_img, 3, 1))], -1).astype(np.float32)
    K = np.array([[61.7, 0, 0.5 * W], [0, 61.7, 0.5 * H], [0, 0, 1]])           # float64, like load_intrinsics (:134-138)
    fake = types.SimpleNamespace(device="cpu", n_imgs=n_img, h=H, w=W, poses=t(poses), images=t(images), K=K, pts0_images=None)
    fake.unravel_idx_from_rayid = lambda ray_id: LLFFDataset.unravel_idx_from_rayid(fake, ray_id)
    ids = rs.randint(0, n_img * H * W, 300).astype(np.int64)
    ids[:4] = [0, n_img * H * W - 1, W - 1, H * W]                            # corners / first pixel of the second image
    out = {"images": images, "pts0": pts0, "poses": poses, "K": K.astype(np.float32), "ids": ids}
    r1 = LLFFDataset.__getitem__(fake, ids.tolist())
    assert "rgbsf_pts0" not in r1
    fake.pts0_images = t(pts0)
    r2 = LLFFDataset.__getitem__(fake, ids.tolist())
    for k, v in r2.items():
        out["out_" + k] = n(v)
        if k in r1:
            assert np.array_equal(n(r1[k]), n(v))
    assert out["out_images_idx"].dtype == np.int64 and out["out_rays_x"].dtype == np.float32
    # add_halfpix=False
    coords = rs.uniform(0, 30, (64, 2)).astype(np.float32)
    c2ws = poses[rs.randint(0, n_img, 64)]
    o, d = get_rays_pix(t(coords), t(K.astype(np.float32)), t(c2ws), add_halfpix=False)
    o2, d2 = get_rays(H, W, t(K.astype(np.float32)), t(poses[1]), add_halfpix=False)
    out.update(nohalf_coords=coords, nohalf_c2ws=c2ws, nohalf_pix_o=n(o), nohalf_pix_d=n(d), nohalf_full_o=n(o2), nohalf_full_d=n(d2))
    save("G28_image_batch", **out)


def G29_pose_track():
    """LLFFEventsDataset.interpolate_poses (data/loader_events.py:133-148) and .sample_events (:259-304) called UNBOUND on an object
    with the attributes they read.  events_pose_bspl is built as load_event_data builds it (:175-182) from the reference's
    _get_slerp_interpolator (utils/data.py:34-62; scipy 1.15.3 here).  Two tracks: 'a' 40 keys, recentred with a c2w from the
    reference's poses_avg, integer coordinates; 'b' 5 keys, no recentring, bd_scale 1, float coordinates (add_halfpix False)."""
    import types
    from data.loader_events import LLFFEventsDataset as EV
    from utils.data import _get_slerp_interpolator, poses_avg
    from utils.events import compute_successor
    from scipy.spatial.transform import Rotation as Rot
    rs = np.random.RandomState(2901)
    Kc = W.synthetic_camera()
    out = {}
    for tag, (M, N, ncoord, nq, recenter, bd_scale, intc) in {"a": (40, 1200, 31 * 17, 200, True, 0.7312, True),
                                                               "b": (5, 300, 13, 64, False, 1.0, False)}.items():
        key_t = np.cumsum(rs.uniform(4e3, 3e4, M)) + 1.7e9
        rv = np.cumsum(rs.standard_normal((M, 3)) * 0.04, 0)
        Rk = Rot.from_rotvec(rv).as_matrix()
        Rk = Rk + rs.standard_normal(Rk.shape) * 2e-8                      # a file's rounding: within the loader's 5e-7 purity check
        Tk = np.cumsum(rs.standard_normal((M, 3)) * 0.03, 0)
        all_poses = np.concatenate([Rk, Tk[..., None]], -1)               # [M, 3, 4] float64 (all_poses_bounds.npy)
        interpolator = _get_slerp_interpolator(key_t, all_poses[:, :3, :3], all_poses[:, :3, 3])

        def events_pose_bspl(tq, key_t=key_t, interpolator=interpolator):   # loader_events.py:176-182
            tq = np.clip(tq, a_min=key_t.min(), a_max=key_t.max())
            irots, itrans = interpolator(tq)
            bottom = np.array([0, 0, 0, 1]).reshape(1, 1, -1).repeat(tq.shape[0], axis=0)
            return np.block([[irots, itrans[..., np.newaxis]], [bottom]]), None
        c2w = None
        if recenter:
            llff = np.concatenate([all_poses[..., 1:2], -all_poses[..., 0:1], all_poses[..., 2:]], -1)
            llff[:, :3, 3] *= bd_scale
            hwf = np.tile(np.array([400.0, 400.0, 400.0]).reshape(1, 3, 1), (M, 1, 1))
            avg = poses_avg(np.concatenate([llff, hwf], -1))              # what recenter_poses(return_c2w=True) computes (utils/data.py:170-172)
            c2w = np.concatenate([avg[:3, :4], np.array([[0, 0, 0, 1.0]])], 0)
        # an event stream inside (and slightly beyond) the key range
        ids = rs.randint(0, ncoord, N).astype(np.int64)
        tms = np.sort(rs.uniform(key_t[0] - 2e3, key_t[-1] + 2e3, N))
        pol = rs.choice([-1.0, 1.0], N)
        ev3 = np.stack([ids, tms, pol], -1)
        succ, nsucc, _, _ = compute_successor(ev3, flat_xy=True)
        events = np.concatenate([ev3, succ.reshape(-1, 1)], -1)
        coords = np.stack([rs.uniform(0, 399, ncoord), rs.uniform(0, 399, ncoord)], -1).astype(np.float32)
        if intc:
            coords = np.floor(coords)
        fake = types.SimpleNamespace(events_pose_bspl=events_pose_bspl, bd_scale=bd_scale, recenter=recenter, recenter_partial=c2w, spherify=False,
                                     events=torch.tensor(events), device="cpu", K=t(Kc), integer_coords=intc, color_events=False,
                                     id_to_coords=t(coords), id_to_color_map=None,
                                     event_accum_min_step=lambda g: 0, event_accum_max_step=lambda g: 0)
        fake.interpolate_poses = lambda tq, fake=fake: EV.interpolate_poses(fake, tq)
        tq = np.concatenate([rs.uniform(key_t[0] - 5e3, key_t[-1] + 5e3, 400), key_t, [key_t[0] - 1.0, key_t[-1] + 1.0]])
        ip = EV.interpolate_poses(fake, tq)
        assert ip.dtype == np.float32 and ip.shape == (tq.shape[0], 4, 4)
        with_succ = np.where(nsucc > 0)[0]
        q = with_succ[rs.randint(0, with_succ.shape[0], nq)].astype(np.int64)
        se = EV.sample_events(fake, torch.tensor(q), 0)
        out.update({f"{tag}_key_t": key_t, f"{tag}_key_poses": all_poses, f"{tag}_bd_scale": np.array(bd_scale), f"{tag}_intc": np.array(int(intc)),
                    f"{tag}_tq": tq, f"{tag}_poses": ip, f"{tag}_events": events, f"{tag}_coords": coords, f"{tag}_ids": q,
                    f"{tag}_rays_start": n(se["events_rays_start"]), f"{tag}_rays_end": n(se["events_rays_end"]),
                    f"{tag}_pos": n(se["events_pos_pol_cumsum"]).astype(np.float32), f"{tag}_neg": n(se["events_neg_pol_cumsum"]).astype(np.float32),
                    f"{tag}_cid": n(se["events_coords_ids"]).astype(np.int64)})
        if c2w is not None:
            out[f"{tag}_recenter_c2w"] = c2w
    save64("G29_pose_track", **out)


def G31_event_hops():
    """The hop schedule of EventsDataset.sample_events (data/loader_events.py:53-70,259-268): utils/misc.py annealing_interpolator (linear /
    cosine / constant) on a grid of steps, and torch_randint_vec (utils/misc.py:87-92) for seeded draws on the CPU generator (torch
    2.10: the fixture pins this version's stream), as sample_events composes them."""
    from utils.misc import annealing_interpolator, torch_randint_vec
    out = {}
    steps = np.array([0, 1, 17, 49, 50, 51, 999, 1000, 5000], dtype=np.int64)
    cases = [(1, 8, 1000), (4, 2, 50), (0, 0, 10), (2, 30, 777)]
    out["steps"], out["cases"] = steps, np.array(cases, dtype=np.int64)
    for m in ("linear", "cosine", "constant"):
        out[f"interp_{m}"] = np.array([[float(annealing_interpolator(a, b, e, m)(int(st))) for st in steps] for a, b, e in cases], dtype=np.float64)
    rs = np.random.RandomState(3101)
    draws = [(1, 8), (2, 30), (1, 1), (3, 3)]
    out["draws"] = np.array(draws, dtype=np.int64)
    for i, (mn, mx) in enumerate(draws):
        ns = torch.tensor(rs.randint(mn + 1, 40, 4096))
        torch.manual_seed(3100 + i)
        hops = torch_randint_vec(torch.tensor(mn) - 1, torch.minimum(torch.tensor(mx), ns) - 1 + 1e-5, torch.int64)      # loader_events.py:265-268
        out[f"ns_{i}"], out[f"hops_{i}"] = ns.numpy().astype(np.int64), hops.numpy().astype(np.int64)
    save64("G31_event_hops", **out)


# --------------------------------------------------------------------------- G32 / G33: the whole training call
def _train_call_model(P=5, n_imgs=6, seed=32):
    """The model run_nerf.py:120-243 builds for the shipped RBK + AWP configs (ViewEmbedding 'param' of width 32, RigidBlurringModel with
    the config's branch sizes, AdaptiveWeightProposal with the config's widths) around the two PDRF levels at PDRF_SMALL grid sizes,
    rgb_add_bias off as shipped.  The blur kernel's rotation / translation heads are scaled from their 1e-5 initialisation to the size a
    trained kernel has (a few 1e-2), the image embeddings drawn normal and the attention logits of the MAM scaled to order 1 (as in G27):
    at the initial values all P rays of a pixel coincide and both softmaxes are uniform, which would pin nothing."""
    from networks.dpnerf.awp import AdaptiveWeightProposal
    from networks.dpnerf.blurmodel import RigidBlurringModel
    from networks.embedding import ViewEmbedding
    from networks.renderer import NeRFAll
    import contextlib
    import io
    torch.manual_seed(seed)
    view_embed = ViewEmbedding(num_embed=n_imgs, embed_dim=32, init_params="normal")
    kern = RigidBlurringModel(feat_ch=0, num_motion=P - 1, D_r=1, W_r=32, D_v=1, W_v=32, D_w=1, W_w=32, output_ch_r=3, output_ch_v=3,
                              rv_window=0.1, use_origin=True, view_embed=view_embed, W=view_embed.out_channels)
    awp = AdaptiveWeightProposal(input_ch=128, num_motion=P - 1, use_origin=True, D_sam=4, W_sam=64, D_mot=1, W_mot=32, dir_freq=2,
                                 rgb_freq=2, depth_freq=3, ray_dir_freq=2, view_feature_ch=view_embed.out_channels)
    awp.load_state_dict({k: t(v) for k, v in W.make_awp_embed_state_dict(seed * 10 + 1).items()}, strict=False)
    rs = np.random.RandomState(seed * 10 + 2)
    bn = awp.MAM.Corr.convd[1]
    with torch.no_grad():
        kern.r_linear.weight.mul_(2e4)
        kern.v_linear.weight.mul_(2e4)
        bn.weight.copy_(t(rs.uniform(0.5, 1.5, 32).astype(np.float32)))
        bn.bias.copy_(t((rs.standard_normal(32) * 0.2).astype(np.float32)))
        bn.running_mean.copy_(t((rs.standard_normal(32) * 0.1).astype(np.float32)))
        bn.running_var.copy_(t(rs.uniform(0.5, 2.0, 32).astype(np.float32)))
        for conv, k in ((awp.MAM.Corr.conva, 8.0), (awp.MAM.Corr.convb, 8.0), (awp.MAM.Corr.convc, 8.0), (awp.MAM.Corr.line_conv_att, 30.0)):
            conv.weight.mul_(k)
    args = ref_import.blurfactory_args(mode="c2f", N_importance=16, N_samples=16, kernel_use_awp=True, kernel_ptnum=P, rgb_add_bias=False, **PDRF_SMALL)
    with contextlib.redirect_stdout(io.StringIO()):
        model = NeRFAll(args, kern, awp)
    gc = [int(v) for v in model.mlp_coarse.gridSize]
    gf = [int(v) for v in model.mlp_fine.gridSize]
    sd = W.make_train_call_state_dict(seed, gc, gf)
    for k in model.state_dict():            # the level parameters only: the kernel and the AWP keep the values set above
        if k.startswith(("mlp_coarse.", "mlp_fine.")) and k not in sd:
            raise KeyError(k)
    model.load_state_dict({k: t(np.asarray(v).copy()) for k, v in sd.items()}, strict=False)
    return model, kern, awp, gc, gf


def _awp_sd(awp):
    return {"awp.sd." + k: n(v).copy() for k, v in awp.state_dict().items() if not k.startswith("sample_feature_embed_layer")}


def _capture_kernel(kern):
    """forward hook on the blur kernel: what it returned (the tensors the GPU-side stub replays), kept in the graph"""
    cap = {}
    def hook(m, i, o):
        cap["new_rays"], cap["weight"], cap["img_embed"] = o[0], o[1], o[3]["img_embed"]
    return cap, kern.register_forward_hook(hook)


TRAIN_CALL_KW = dict(retraw=True, perturb=0., N_importance=16, N_samples=16, use_viewdirs=True, white_bkgd=False, raw_noise_std=0., inference=False)


def G32_train_forward():
    """NeRFAll.forward IN TRAINING MODE as run_nerf.py:438-442 calls it behind kernel_start_iter (networks/renderer.py:277-392): the real
    RigidBlurringModel (dpnerf/blurmodel.py:129-173) -> render of the P warped rays per pixel (mode='c2f') -> the real
    AdaptiveWeightProposal (dpnerf/awp.py:79-117) on depth_feature / z_vals / the NDC rays_d / img_embed -> ccw_fine + ccw_fine * 0.05
    renormalised -> the two rbk_weighted_sums -> TV x 5 -> other_tensors.  32 pixels x P = 5, 16 + 16 samples, perturb = 0, raw noise 0.
    Recorded: the kernel's outputs (new_rays, weight, img_embed: a stub kernelsnet replays them where the reference cannot travel), every
    output of the call, and torch.autograd gradients of a fixed projection of ALL outputs w.r.t. every level parameter (summaries +
    element pins), every AWP parameter, new_rays, weight and img_embed (full; img_embed's gradient both as it arrives through the AWP and in
    total, i.e. with the part through the kernel's own branches)."""
    P, R = 5, 32
    model, kern, awp, gc, gf = _train_call_model(P)
    model.train(True)
    K = W.synthetic_camera()
    rays = W.synthetic_rays(32, R)
    rs = np.random.RandomState(3201)
    images_idx = rs.randint(0, 6, (R, 1)).astype(np.int64)
    proj = {k: rs.standard_normal((R, 3)).astype(np.float32) for k in ("rgb", "rgb1", "rgb_awp", "stage1_rgb_pts0", "stage1_rgb1_pts0")}
    awp_before = _awp_sd(awp)
    cap, hk = _capture_kernel(kern)
    seen = {}
    hk2 = awp.register_forward_hook(lambda m, i, o: seen.update(z_vals=n(i[1]).copy(), rays_d=n(i[2]).copy(), awp_out=n(o).copy()))
    # the kernel's img_embed output IS its own branches' input (blurmodel.py:133-136,171): an identity node in front of the AWP separates the
    # gradient that arrives through the AWP (what a replayed kernel output can receive) from the total
    hk3 = awp.register_forward_pre_hook(lambda m, i: (seen.update(vf_awp=i[3] * 1.0), (i[0], i[1], i[2], seen["vf_awp"]))[1])
    with torch.enable_grad():
        rgb, rgb1, other_loss, other_tensors = model(400, 400, t(K), chunk=1 << 20, rays=t(rays), rays_info={"images_idx": t(images_idx)},
                                                     force_naive=False, return_pts0_rgb=True, **TRAIN_CALL_KW)
        outs = dict(rgb=rgb, rgb1=rgb1, rgb_awp=other_tensors["rgb_awp"], stage1_rgb_pts0=other_tensors["stage1_rgb_pts0"],
                    stage1_rgb1_pts0=other_tensors["stage1_rgb1_pts0"])
        tv = other_loss["TV"]
        loss = sum((outs[k] * t(proj[k])).sum() for k in proj) + 0.1 * tv.sum()
        lv = [(k, p) for k, p in model.named_parameters() if k.startswith(("mlp_coarse.", "mlp_fine."))]
        ap = [(k, p) for k, p in awp.named_parameters() if not k.startswith("MAM.conv.")]
        kp = list(kern.named_parameters())
        grads = torch.autograd.grad(loss, [p for _, p in lv] + [p for _, p in ap] + [cap["new_rays"], cap["weight"], seen["vf_awp"]] + [p for _, p in kp] +
                                    [cap["img_embed"]], allow_unused=True)
    hk.remove()
    hk2.remove()
    hk3.remove()
    g_img_embed_total, grads = grads[-1], grads[:-1]
    assert set(other_loss) == {"TV"} and set(other_tensors) == {"rgb_awp", "stage1_img_embed", "stage1_rgb_pts0", "stage1_rgb1_pts0"}, (set(other_loss), set(other_tensors))
    out = {}
    _grad_summaries([(k, g) for (k, _), g in zip(lv, grads) if g is not None], out, "g.")
    o = len(lv)
    for (k, _), g in zip(ap, grads[o:o + len(ap)]):
        if g is not None:
            out["g.awp." + k] = n(g)
    o += len(ap)
    out["g.new_rays"], out["g.weight"], out["g.img_embed"] = n(grads[o]), n(grads[o + 1]), n(grads[o + 2])       # (img_embed: through the AWP)
    out["g.img_embed_total"] = n(g_img_embed_total)                                                                 # (+ through the kernel's own branches)
    for (k, _), g in zip(kp, grads[o + 3:]):            # (for a maintainer who runs the real kernel module: what its parameters receive)
        if g is not None:
            out["g.kernel." + k] = n(g)
    bn = awp.MAM.Corr.convd[1]
    for k in ("running_mean", "running_var", "num_batches_tracked"):
        out["awp.after." + k] = n(getattr(bn, k))
    save("G32_train_forward", rays=rays, images_idx=images_idx, new_rays=n(cap["new_rays"]), weight=n(cap["weight"]), img_embed=n(cap["img_embed"]),
         **{"out." + k: n(v) for k, v in outs.items()}, **{"proj." + k: v for k, v in proj.items()}, tv=np.float64(tv.sum().item()),
         loss=np.float64(loss.item()), stage1_img_embed=n(other_tensors["stage1_img_embed"]), awp_in_z_vals=seen["z_vals"], awp_in_rays_d=seen["rays_d"],
         awp_out=seen["awp_out"], grid_coarse=np.array(gc), grid_fine=np.array(gf),
         **awp_before, **out)


def G33_train_trajectory():
    """K = 5 iterations of the reference's optimisation loop on the G32 model (run_nerf.py:423-613 with the blurfactory config's switches:
    blur batch through kernel + AWP, loss = (img + img0) (1 - flw) + img_awp flw + w_pts0 (pts0 terms) + TV + w_egm event_egm with the event
    batch's start / end rays rendered force_naive=True, gamma CRF on colours, learnable event-CRF with the 'pos-neg' features; Adam
    betas (0.9, 0.999) over the groups of run_nerf.py:252-263 {grad_vars, grad_vars_vol, crf} + the exponential decay of :603-613 with
    lrate_decay 10).  lrate is 5e-4, not the config's 5e-3: five steps at 5e-3 move the 64-wide coarse networks by their own size.
    Recorded per step: the kernel's outputs (replayed by the stub on the GPU side -- the kernel's own parameters are trained by the
    reference loop and not compared), the loss, and (norm, projection, element pins) of every level / AWP / CRF parameter AFTER the step."""
    from utils.events import egm_loss
    sys.modules.setdefault("skimage", type(sys)("skimage"))
    sys.modules["skimage"].metrics = None
    sys.modules.setdefault("networks.lpips", type(sys)("networks.lpips"))
    sys.modules["networks.lpips"].LPIPS = None
    from utils.metrics import img2mse
    P, R, NE, STEPS = 5, 32, 48, 5
    lrate, lrate_decay, flw, w_pts0, w_egm, w_tv, thr = 5e-4, 10, 0.1, 0.01, 0.1, 1.0, 0.2
    model, kern, awp, gc, gf = _train_call_model(P, seed=33)
    tm = _tonemap("gamma", "learn", 2, 331)
    for prm in tm.tonemapping_event.parameters():
        if prm.dim() == 2:
            prm.data.mul_(3.0)
    K = W.synthetic_camera()
    rs = np.random.RandomState(3301)
    rays = W.synthetic_rays(33, R)
    ev_start, ev_end = W.synthetic_rays(34, NE), W.synthetic_rays(35, NE)
    images_idx = rs.randint(0, 6, (R, 1)).astype(np.int64)
    target = rs.uniform(0, 1, (R, 3)).astype(np.float32)
    target_pts0 = rs.uniform(0, 1, (R, 3)).astype(np.float32)
    cn = -rs.randint(0, 4, NE).astype(np.float32)
    cp = rs.randint(0, 4, NE).astype(np.float32)
    awp_before = _awp_sd(awp)
    optim_params = [{"params": model.grad_vars, "lr": lrate}, {"params": model.grad_vars_vol, "lr": lrate}, {"params": tm.parameters(), "lr": lrate}]
    for g in optim_params:
        g.setdefault("initial_lr", g["lr"])
    opt = torch.optim.Adam(params=optim_params, lr=lrate, betas=(0.9, 0.999))
    crf = lambda x, **k: tm(x, **k)
    cap, hk = _capture_kernel(kern)
    out = {}
    losses = []
    model.train(True)
    tm.train()
    tracked = lambda: ([(k, p) for k, p in model.named_parameters() if k.startswith(("mlp_coarse.", "mlp_fine."))] +
                       [("awp." + k, p) for k, p in awp.named_parameters() if not k.startswith("MAM.conv.")] +
                       [("crf." + k, p) for k, p in tm.tonemapping_event.named_parameters()])
    p0 = {k: n(p).copy() for k, p in tracked()}
    global_step = 0
    for i in range(STEPS):
        with torch.enable_grad():
            rgb, rgb0, extra_loss, extra_tensor = model(400, 400, t(K), chunk=1 << 20, rays=t(rays), rays_info={"images_idx": t(images_idx)},
                                                        force_naive=False, return_pts0_rgb=True, **TRAIN_CALL_KW)
            rgb, rgb0 = crf(rgb, mode="encode_rgb"), crf(rgb0, mode="encode_rgb")
            loss = img2mse(rgb, t(target)) + img2mse(rgb0, t(target))
            img_fine = img2mse(crf(extra_tensor["rgb_awp"], mode="encode_rgb"), t(target))
            loss = loss * (1 - flw) + img_fine * flw
            pts0 = 0.0
            for name in ("stage0_rgb_pts0", "stage1_rgb_pts0", "stage1_rgb1_pts0"):
                if name in extra_tensor:
                    pts0 = pts0 + img2mse(crf(extra_tensor[name], mode="encode_rgb"), t(target_pts0))
            loss = loss + pts0 * w_pts0
            extra_loss.update({k: torch.mean(v) for k, v in extra_loss.items()})
            loss = loss + extra_loss["TV"] * w_tv
            bii = (t(np.array([thr, thr], np.float32)) * torch.stack([t(cn), t(cp)], -1)).sum(-1)
            feat = torch.stack([t(cn), t(cp)], -1)
            s, s0, _, _ = model(400, 400, t(K), chunk=1 << 20, rays=t(ev_start), rays_info=None, force_naive=True, **TRAIN_CALL_KW)
            e, e0, _, _ = model(400, 400, t(K), chunk=1 << 20, rays=t(ev_end), rays_info=None, force_naive=True, **TRAIN_CALL_KW)
            lum = lambda x: crf(x, mode="encode_luma", ev_extra_feat=feat)
            egm = egm_loss(lum(s0), lum(e0), bii, color_mask=None, color_weight=None) + egm_loss(lum(s), lum(e), bii, color_mask=None, color_weight=None)
            loss = loss + egm * w_egm
            opt.zero_grad()
            loss.backward()
        opt.step()
        decay_rate, decay_steps = 0.1, lrate_decay * 1000
        for g in opt.param_groups:
            g["lr"] = g["initial_lr"] * (decay_rate ** (global_step / decay_steps))
        global_step += 1
        losses.append(loss.item())
        out[f"s{i}.new_rays"], out[f"s{i}.weight"], out[f"s{i}.img_embed"] = n(cap["new_rays"]).copy(), n(cap["weight"]).copy(), n(cap["img_embed"]).copy()
        _grad_summaries([(k, t(n(p) - p0[k])) for k, p in tracked()], out, f"s{i}.d.", elements=i in (0, STEPS - 1))
        for k, p in tracked():              # small tensors in full (AWP per-ray layers, CRF)
            if p.numel() <= 4096 and i in (0, STEPS - 1):
                out[f"s{i}.p.{k}"] = n(p).copy()
    hk.remove()
    bn = awp.MAM.Corr.convd[1]
    for k in ("running_mean", "running_var", "num_batches_tracked"):
        out["awp.after." + k] = n(getattr(bn, k))
    print("   losses:", ["%.6f" % v for v in losses])
    save("G33_train_trajectory", rays=rays, ev_start=ev_start, ev_end=ev_end, images_idx=images_idx, target=target, target_pts0=target_pts0, cn=cn, cp=cp,
         losses=np.array(losses, np.float64), scalars=np.array([lrate, lrate_decay, flw, w_pts0, w_egm, w_tv, thr], np.float64),
         grid_coarse=np.array(gc), grid_fine=np.array(gf), **awp_before, **out)


# --------------------------------------------------------------------------- G34 / G35: the once-per-dataset tables of the loaders
def _patched_numpy_for_reference():
    """The reference was written for numpy < 1.24 (np.bool, and np.unique's 1-D inverse for an [M, 1] void view): both restored for the
    duration of a call.  Nothing of the reference's arithmetic is touched."""
    import contextlib

    @contextlib.contextmanager
    def ctx():
        had_bool = hasattr(np, "bool")
        old_unique = np.unique
        if not had_bool:
            np.bool = bool

        def unique(*a, **k):
            r = old_unique(*a, **k)
            if k.get("return_inverse") and isinstance(r, tuple):
                r = tuple(x.reshape(-1) if i == (2 if k.get("return_index") else 1) else x for i, x in enumerate(r))
            return r
        np.unique = unique
        try:
            yield
        finally:
            np.unique = old_unique
            if not had_bool:
                del np.bool
    return ctx()


def G34_event_tables():
    """LLFFEventsDataset.load_event_data (data/loader_events.py:150-257, with utils/events.py:11-69 load_events_h5 and compute_successor)
    called UNBOUND on an object with the attributes it reads.  The files it opens are replaced by arrays IN THIS GENERATOR ONLY: np.load is
    dispatched on the file name (timestamps.npz, all_timestamps.npy, all_poses_bounds.npy, ev_map.npz), h5py.File returns the x / y / t / p
    arrays, os.path.exists answers for ev_map.npz.  Two streams: 'int' -- integer pixel coordinates on a 12 x 16 sensor, polarities 0 / 1
    (normalised to -1 / 1), events before and after the pose range (filtered), colour events with the Bayer pattern, single-hop successor
    filter; 'flt' -- rectified float coordinates with the ev_map inverse maps (the id_to_color_map loop of :219-236), multi-hop
    accumulation range [1, 3] -> [2, 4] (events_with_successor_idx keeps num_successors > 2)."""
    import types
    import data.loader_events as LE
    import utils.events as UE
    from scipy.spatial.transform import Rotation as Rot
    rs = np.random.RandomState(3401)
    out = {}
    for tag, (h, w, N, M, flt, acc, acc_end) in {"int": (12, 16, 700, 9, False, [0, 0], [0, 0]), "flt": (10, 14, 500, 7, True, [1, 3], [2, 4])}.items():
        key_t = (np.cumsum(rs.randint(4000, 30000, M)) + 100000).astype(np.int64)          # all_timestamps.npy, microseconds
        rv = np.cumsum(rs.standard_normal((M, 3)) * 0.04, 0)
        Rk = Rot.from_rotvec(rv).as_matrix()
        Tk = np.cumsum(rs.standard_normal((M, 3)) * 0.03, 0)
        p35 = np.concatenate([Rk, Tk[..., None], np.tile(np.array([h, w, 20.0]).reshape(1, 3, 1), (M, 1, 1))], -1)
        apb = np.concatenate([p35.reshape(M, 15), rs.uniform(0.5, 4.0, (M, 2))], -1)                 # all_poses_bounds.npy [M, 17]
        n_img = 3
        it = np.linspace(key_t[1], key_t[-2], n_img)
        tms_npz = {"timestamps": it, "timestamps_start": it - 1500.0, "timestamps_end": it + 1500.0}
        # the stream: most pixels active, a few silent; timestamps partly outside the pose range
        act = rs.rand(h, w) < 0.85
        ys, xs = np.where(act)
        pick = rs.randint(0, ys.shape[0], N)
        px, py = xs[pick].astype(np.float32), ys[pick].astype(np.float32)
        if flt:                                       # a smooth rectification of the pixel grid: the coordinates the events carry
            rect = lambda X, Y: ((X + 0.31 * np.sin(0.4 * Y) + 0.25).astype(np.float32), (Y + 0.27 * np.cos(0.3 * X) - 0.125).astype(np.float32))
            ex, ey = rect(px, py)
            gx, gy = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
            inv_mapx, inv_mapy = rect(gx, gy)
            ev_map = {"inv_mapx": inv_mapx, "inv_mapy": inv_mapy}
        else:
            ex, ey, ev_map = px, py, None
        et = np.sort(rs.randint(key_t[0] - 9000, key_t[-1] + 9000, N)).astype(np.int64)
        ep = rs.randint(0, 2, N).astype(np.int64)                                                    # 0 / 1 in the file

        def np_load(path, *a, _k=key_t, _apb=apb, _t=tms_npz, _m=ev_map, **kw):
            name = os.path.basename(path)
            return {"timestamps.npz": _t, "all_timestamps.npy": _k.copy(), "all_poses_bounds.npy": _apb.copy(), "ev_map.npz": _m}[name]
        fake = types.SimpleNamespace(events_tms_files_unit="us", events_tms_unit="us", basedir="/nowhere", h=h, w=w, color_events=True,
                                     event_accumulate_step_range=acc, event_accumulate_step_range_end=acc_end)
        real_load, real_exists, real_h5 = np.load, os.path.exists, sys.modules["h5py"]
        try:
            np.load = np_load
            os.path.exists = lambda pth, _f=flt: _f if pth.endswith("ev_map.npz") else real_exists(pth)
            h5 = types.ModuleType("h5py")
            h5.File = lambda path, mode="r", _d={"x": ex, "y": ey, "t": et, "p": ep}: {k: v.copy() for k, v in _d.items()}
            sys.modules["h5py"] = UE.h5py = h5
            import io
            import contextlib
            with _patched_numpy_for_reference(), contextlib.redirect_stdout(io.StringIO()):
                ret = LE.LLFFEventsDataset.load_event_data(fake)
        finally:
            np.load, os.path.exists = real_load, real_exists
            sys.modules["h5py"] = UE.h5py = real_h5
        assert bool(ret["intcoords"]) == (not flt)
        c2i = ret["coords_to_id"]
        if flt:                                      # the dict of :201: keys (x, y) -> id, stored as arrays
            c2i = np.array([[k[0], k[1], v] for k, v in c2i.items()], dtype=np.float64)
        out.update({f"{tag}_x": ex, f"{tag}_y": ey, f"{tag}_t": et, f"{tag}_p": ep, f"{tag}_hw": np.array([h, w]), f"{tag}_key_t": key_t, f"{tag}_apb": apb,
                    f"{tag}_acc": np.array(acc + acc_end), f"{tag}_img_t": it,
                    f"{tag}_events": np.asarray(ret["events"]).astype(np.float64), f"{tag}_id_to_coords": np.asarray(ret["id_to_coords"]).astype(np.float64),
                    f"{tag}_id_to_color_map": np.asarray(ret["id_to_color_map"]).astype(np.uint8), f"{tag}_coords_to_id": np.asarray(c2i),
                    f"{tag}_num_successors": np.asarray(ret["events_num_successors"]).astype(np.int64),
                    f"{tag}_with_successor_idx": np.asarray(ret["events_with_successor_idx"]).astype(np.int64),
                    f"{tag}_allknown_poses": np.asarray(ret["allknown_poses"]).astype(np.float64)})
        if flt:
            out.update({f"{tag}_inv_mapx": inv_mapx, f"{tag}_inv_mapy": inv_mapy})
        tq = rs.uniform(key_t[0] - 100, key_t[-1] + 100, 40)
        ip, _ = ret["events_pose_bspl"](tq)
        out[f"{tag}_tq"], out[f"{tag}_pose_bspl"] = tq, np.asarray(ip).astype(np.float64)
        print(f"   {tag}: {np.asarray(ret['events']).shape[0]} of {N} events inside the pose range, {np.asarray(ret['id_to_coords']).shape[0]} coordinate ids, "
              f"{np.asarray(ret['events_with_successor_idx']).shape[0]} events with enough successors")
    save64("G34_event_tables", **out)


def G35_llff_poses():
    """LLFFDataset.load_poses (data/loader.py:178-203) + the recentring branch of recenter_spherify_poses (:205-216, utils/data.py:167-183
    recenter_poses / poses_avg) called UNBOUND, np.load replaced by the array: poses_bounds [N, 17] -> LLFF column change, float32 cast,
    bd_factor rescale of translations and bounds, recentred poses + the average pose (recenter_partial, what the event loader re-applies)."""
    import types
    _stub_voxels()
    import data.loader as DL
    from utils.data import recenter_poses
    from scipy.spatial.transform import Rotation as Rot
    rs = np.random.RandomState(3501)
    out = {}
    for tag, (N, factor, bd_factor) in {"a": (7, 1, 0.75), "b": (4, 2, None)}.items():
        Rk = Rot.from_rotvec(rs.standard_normal((N, 3)) * 0.2).as_matrix()
        Tk = rs.standard_normal((N, 3)) * 0.5
        p35 = np.concatenate([Rk, Tk[..., None], np.tile(np.array([600.0, 800.0, 415.0]).reshape(1, 3, 1), (N, 1, 1))], -1)
        pb = np.concatenate([p35.reshape(N, 15), rs.uniform(1.5, 9.0, (N, 2))], -1)
        fake = types.SimpleNamespace(basedir="/nowhere")
        real_load = np.load
        try:
            np.load = lambda path, *a, _pb=pb, **k: _pb.copy()
            poses, bds, sc = DL.LLFFDataset.load_poses(fake, factor, (600 // factor, 800 // factor, 3), bd_factor=bd_factor)
        finally:
            np.load = real_load
        rec, c2w = recenter_poses(poses, return_c2w=True)
        out.update({f"{tag}_poses_bounds": pb, f"{tag}_args": np.array([factor, -1.0 if bd_factor is None else bd_factor, 600 // factor, 800 // factor]),
                    f"{tag}_poses": np.asarray(poses), f"{tag}_bds": np.asarray(bds), f"{tag}_sc": np.array(sc, dtype=np.float64),
                    f"{tag}_recentered": np.asarray(rec), f"{tag}_c2w": np.asarray(c2w).astype(np.float64)})
    save64("G35_llff_poses", **out)


ALL = [G1_embedder, G2_nerf_mlp, G3_nerf_raw2outputs, G4_voxel_raw2outputs, G5_sample_pdf, G6_rays,
       G7_render_nerf, G8_appfeature, G9_render_c2f, G10_rbk_weighted_sum, G11_crf, G12_egm_loss, G13_edi,
       G14_loss_assembly, G15_awp_feature_integration, G16_rbk_warp, G17_compute_successor, G18_nerf_grads, G19_c2f_grads, G20_loss_grads,
       G21_awp_sample_embed, G22_mam, G23_render_nerf_no_viewdirs, G24_render_other_multires, G25_pbe_composite_feature,
       G26_sample_events, G27_awp_per_ray, G28_image_batch, G29_pose_track, G30_c2f_grads_16k, G31_event_hops, G32_train_forward,
       G33_train_trajectory, G34_event_tables, G35_llff_poses]

if __name__ == "__main__":
    want = set(sys.argv[1:])
    for fn in ALL:
        tag = fn.__name__.split("_")[0]
        if want and tag not in want:
            continue
        print(fn.__name__)
        fn()
