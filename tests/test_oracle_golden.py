"""Pins the CPU oracle (oracle/evd_oracle.c) against golden vectors produced by the imported
reference (tools/gen_golden.py). CPU only. Tolerances: the oracle and torch-CPU differ only in
summation order and libm vs SLEEF transcendentals, so 2e-6 absolute on O(1) values."""
import numpy as np
import pytest

from conftest import load_golden, maxabs, sample_pdf_flip_report, train_call_errors, z_mismatch
from evdeblurnerf_amd import weights as W
from oracle import oracle as O

TOL = 2e-6


def test_G1_embedder():
    g = load_golden("G1_embedder")
    for L, key in ((10, "pe10"), (4, "pe4"), (2, "pe2")):
        out = O.embed(g["x"], L)
        # |arg| reaches 40*512: sinf/cosf of both sides are correctly rounded to ~1ulp of the result
        assert maxabs(out, g[key]) < TOL, key


@pytest.mark.parametrize("tag,seed,Wd,bias", [("w256", 7, 256, True), ("w256_nobias", 8, 256, False), ("w64", 9, 64, True)])
def test_G2_nerf_mlp(tag, seed, Wd, bias):
    g = load_golden("G2_nerf_mlp")
    sd = W.make_nerf_state_dict(seed, W=Wd, rgb_add_bias=bias)
    net = O.Nerf(sd, W=Wd)
    emb = np.concatenate([O.embed(g["pts"], 10), O.embed(g["dirs"], 4)], -1)
    raw, fa, fb = O.nerf_mlp(net, emb, want_after=True, want_before=True)
    assert maxabs(raw, g[f"raw_{tag}"]) < 5e-6
    assert maxabs(fa[:, :16], g[f"feat_{tag}"]) < 5e-6
    assert maxabs(fb[:, :16], g[f"featb_{tag}"]) < 5e-6


@pytest.mark.parametrize("S", [64, 128, 33])
def test_G3_nerf_raw2outputs(S):
    g = load_golden("G3_nerf_raw2outputs")
    raw, z, d = g[f"raw_S{S}"], g[f"z_S{S}"], g[f"d_S{S}"]
    cases = {"plain": {}, "white": dict(white_bkgd=True), "rmnear": dict(rmnear_thresh=20 / 128),
             "relu_rgb": dict(rgb_act="relu"), "none_rgb": dict(rgb_act="none"), "softplus": dict(sigma_act="softplus")}
    for tag, kw in cases.items():
        o = O.composite(raw, z, d, **kw)
        for k in ("rgb", "acc", "depth", "weights"):
            assert maxabs(o[k], g[f"{k}_S{S}_{tag}"]) < 5e-6, (tag, k)
        if tag == "plain":
            assert maxabs(o["density"], g[f"density_S{S}"]) < TOL
            # the last alpha is forced to 1 => acc == 1 up to rounding (SURVEY "read this first" #4)
            assert np.allclose(o["acc"], 1.0, atol=1e-5)
    o = O.composite(raw, z, d, feature=g[f"feat_S{S}"])
    assert maxabs(o["fmap"], g[f"fmap_S{S}"]) < 5e-6


@pytest.mark.parametrize("S", [64, 128])
def test_G4_voxel_raw2outputs(S):
    g = load_golden("G4_voxel_raw2outputs")
    raw, z, d = g[f"raw_S{S}"], g[f"z_S{S}"], g[f"d_S{S}"]
    for tag, act in (("coarse", "relu"), ("fine", "none")):
        o = O.composite(raw, z, d, sigma_ch=0, rgb_ch0=1, n_rgb=3, rgb_act=act)
        for k in ("rgb", "acc", "depth", "weights"):
            assert maxabs(o[k], g[f"{k}_S{S}_{tag}"]) < 5e-6, (tag, k)
    o = O.composite(g[f"raw16_S{S}"], z, d, sigma_ch=0, rgb_ch0=1, n_rgb=15, rgb_act="relu")
    assert maxabs(o["rgb"], g[f"fmap16_S{S}"]) < 5e-6


@pytest.mark.parametrize("S,N", [(64, 64), (64, 128), (128, 64), (17, 9)])
def test_G5_sample_pdf(S, N):
    g = load_golden("G5_sample_pdf")
    key = f"S{S}_N{N}"
    bins, w, u = g[f"bins_{key}"], g[f"w_{key}"], g[f"u_{key}"]
    ulin = np.linspace(0.0, 1.0, N).astype(np.float32)
    det = O.sample_pdf(bins, w, N, det=True)
    nbad, unexplained = sample_pdf_flip_report(det, g[f"det_{key}"], bins, w, ulin)
    assert unexplained == 0 and nbad <= 0.01 * det.size, (nbad, unexplained)
    rnd = O.sample_pdf(bins, w, N, det=False, u=u)
    nbad, unexplained = sample_pdf_flip_report(rnd, g[f"rand_{key}"], bins, w, u)
    assert unexplained == 0 and nbad <= 0.005 * rnd.size, (nbad, unexplained)
    # everything that is not ill-conditioned agrees to rounding
    assert (np.abs(det - g[f"det_{key}"]) <= 5e-6).mean() > 0.99


def test_G6_rays():
    g = load_golden("G6_rays")
    o, d = O.get_rays(60, 80, g["Kn"], g["c2w"])
    assert maxabs(o[::7, ::5], g["rays_o_full"]) == 0.0
    assert maxabs(d[::7, ::5], g["rays_d_full"]) < 1e-6
    K = W.synthetic_camera()
    op, dp = O.get_rays_pix(g["coords"], K, g["poses"])
    assert maxabs(op, g["rays_o_pix"]) == 0.0
    assert maxabs(dp, g["rays_d_pix"]) < 1e-6
    on, dn = O.ndc_rays(400, 400, float(K[0, 0]), 1.0, g["rays_o_pix"], g["rays_d_pix"])
    assert maxabs(on, g["ndc_o"]) < 2e-6
    assert maxabs(dn, g["ndc_d"]) < 2e-6


def _check_render(res, g, prefix, keys, tol):
    for k_out, k_g in keys.items():
        assert maxabs(res[k_out], g[prefix + k_g]) < tol, k_out


def test_G7_render_nerf():
    g = load_golden("G7_render_nerf")
    c = O.Nerf(W.make_nerf_state_dict(11))
    f = O.Nerf(W.make_nerf_state_dict(12))
    keys_h = dict(rgb="rgb", depth="depth", acc="acc", z_vals="z_vals", weights="weights", rgb0="rgb0",
                  depth0="depth0", acc0="acc0", z_std="z_std", z_vals0="z_vals0", weights0="weights0")
    res = O.render_nerf(c, f, O.make_cfg(N_samples=64, N_importance=64), W.synthetic_rays(1, 96))
    _check_render(res, g, "a_", keys_h, 2e-5)
    m = O.Nerf(W.make_nerf_state_dict(13))
    res = O.render_nerf(m, None, O.make_cfg(N_samples=128), W.synthetic_rays(2, 80))
    _check_render(res, g, "b_", dict(rgb="rgb", depth="depth", acc="acc", z_vals="z_vals", weights="weights"), 2e-5)
    res = O.render_nerf(m, None, O.make_cfg(N_samples=64, ndc=False, near=0.5, far=3.5, white_bkgd=True, lindisp=True),
                        W.synthetic_rays(3, 64))
    _check_render(res, g, "c_", dict(rgb="rgb", depth="depth", acc="acc"), 2e-5)
    res = O.render_nerf(c, f, O.make_cfg(N_samples=64, N_importance=32, perturb=1.0), W.synthetic_rays(4, 48),
                        t_rand=g["d_t_rand"], u=g["d_u"])
    _check_render(res, g, "d_", keys_h, 2e-5)


def test_G23_render_nerf_without_viewdirs():
    """use_viewdirs=False (renderer.py:443-446: 8-column ray batch; nerf.py:41-44,158-160: output_linear head, output_ch 5 with importance
    sampling) vs the golden the reference's NeRFAll produced; the per-sample feature is the "before_linear" one."""
    g = load_golden("G23_render_nerf_no_viewdirs")
    mk = lambda seed, och: O.Nerf(W.make_nerf_state_dict(seed, input_ch_views=0, use_viewdirs=False, output_ch=och), input_ch_views=0,
                                  use_viewdirs=False, output_ch=och)
    keys_h = dict(rgb="rgb", depth="depth", acc="acc", z_vals="z_vals", weights="weights", rgb0="rgb0",
                  depth0="depth0", acc0="acc0", z_std="z_std", z_vals0="z_vals0", weights0="weights0")
    rays = W.synthetic_rays(23 + 32, 72)
    cfg = O.make_cfg(N_samples=48, N_importance=32, use_viewdirs=False)
    res = O.render_nerf(mk(61, 5), mk(62, 5), cfg, rays)
    _check_render(res, g, "a_", keys_h, 2e-5)
    feat = O.render_nerf(mk(61, 5), mk(62, 5), cfg, rays[:2], want_feature=True)["feature"]
    assert maxabs(feat, g["a_depth_feature2"]) < 2e-5
    rays = W.synthetic_rays(23, 40)
    cfg = O.make_cfg(N_samples=128, use_viewdirs=False, ndc=False, near=0.5, far=3.5)
    res = O.render_nerf(mk(61, 4), None, cfg, rays)
    _check_render(res, g, "b_", dict(rgb="rgb", depth="depth", acc="acc", z_vals="z_vals", weights="weights"), 2e-5)
    assert maxabs(O.render_nerf(mk(61, 4), None, cfg, rays[:2], want_feature=True)["feature"], g["b_depth_feature2"]) < 2e-5


AABB = [-1.5, -1.5, -1.0, 1.5, 1.5, 1.0]


def _voxels(seed_c, seed_f):
    gc = W.pdrf_grid_size(AABB[:3], AABB[3:], 24 ** 3)
    gf = W.pdrf_grid_size(AABB[:3], AABB[3:], 48 ** 3)
    sdc = W.make_pdrf_state_dict(seed_c, gc, input_ch=95, hidden_dim=64, geo_feat_dim=15)
    sdf = W.make_pdrf_state_dict(seed_f, gf, input_ch=127, hidden_dim=256, geo_feat_dim=128)
    vc = O.Voxel(sdc, "", gc, AABB, input_ch=95, hidden_dim=64, geo_feat_dim=15, rgb_act="relu")
    vf = O.Voxel(sdf, "", gf, AABB, input_ch=127, hidden_dim=256, geo_feat_dim=128, rgb_act="none")
    return vc, vf, gc, gf


def test_G8_appfeature():
    g = load_golden("G8_appfeature")
    vc, vf, gc, gf = _voxels(21, 22)
    assert gc == list(g["grid_coarse"]) and gf == list(g["grid_fine"])
    pts = g["pts"]
    assert maxabs(O.appfeature(vc, pts).reshape(g["ft_coarse"].shape), g["ft_coarse"]) < 2e-6
    assert maxabs(O.appfeature(vf, pts).reshape(g["ft_fine"].shape), g["ft_fine"]) < 2e-6


def test_G9_render_c2f():
    g = load_golden("G9_render_c2f")
    vc, vf, _, _ = _voxels(31, 32)
    rays = W.synthetic_rays(9, 64)
    res = O.render_c2f(vc, vf, O.make_cfg(N_samples=64, N_importance=64), rays, want_feature=True)
    # c2f carries the tri-plane gather + two MLP levels: rounding-order noise reaches a few 1e-5 on depth
    for k in ("rgb", "depth", "acc", "rgb0", "depth0", "acc0", "z_std", "z_vals0", "weights0"):
        assert maxabs(res[k], g[k]) < 5e-5, k
    frac, worst = z_mismatch(res["z_vals"], g["z_vals"])
    assert frac < 0.005 and worst < 1.0 / 63 + 1e-4, (frac, worst)
    same = np.abs(res["z_vals"] - g["z_vals"]).max(-1) < 5e-5          # rays whose sample sets agree
    assert same.mean() > 0.8
    assert maxabs(res["weights"][same], g["weights"][same]) < 5e-5
    tight = np.abs(res["z_vals"] - g["z_vals"]).max(-1)[:16] < 2e-6   # features move ~20x the sample offset
    assert tight.sum() >= 4
    assert maxabs(res["feature"][:16, :, :8][tight], g["f_depth_feature"][tight]) < 1e-4
    res = O.render_c2f(vc, None, O.make_cfg(N_samples=64, N_importance=0), rays)
    for k in ("rgb", "depth", "acc", "weights"):
        assert maxabs(res[k], g["c_" + k]) < 2e-5, k
    cfg = O.make_cfg(N_samples=64, N_importance=64)
    assert maxabs(O.ray_batch(cfg, rays[:16])[:, 3:6], g["f_rays_d"]) < 2e-6


def test_G24_render_other_multires():
    """frequency counts other than (10, 4) (options.py:94-97; embedding.py:101-117; renderer.py:18,44), nerf and c2f modes"""
    g = load_golden("G24_render_other_multires")
    keys_h = dict(rgb="rgb", depth="depth", acc="acc", z_vals="z_vals", weights="weights", rgb0="rgb0",
                  depth0="depth0", acc0="acc0", z_std="z_std", z_vals0="z_vals0", weights0="weights0")
    L, Lv = 6, 2
    mk = lambda seed: O.Nerf(W.make_nerf_state_dict(seed, input_ch=W.pe_dim(L), input_ch_views=W.pe_dim(Lv)), input_ch=W.pe_dim(L),
                             input_ch_views=W.pe_dim(Lv))
    res = O.render_nerf(mk(71), mk(72), O.make_cfg(N_samples=48, N_importance=32, multires=L, multires_views=Lv), W.synthetic_rays(41, 56))
    _check_render(res, g, "a_", keys_h, 2e-5)
    L, Lv = 3, 8
    m = O.Nerf(W.make_nerf_state_dict(73, D=4, W=64, input_ch=W.pe_dim(L), input_ch_views=W.pe_dim(Lv), skips=()), D=4, W=64,
               input_ch=W.pe_dim(L), input_ch_views=W.pe_dim(Lv))
    res = O.render_nerf(m, None, O.make_cfg(N_samples=64, ndc=False, near=0.5, far=3.5, multires=L, multires_views=Lv), W.synthetic_rays(42, 40))
    _check_render(res, g, "b_", dict(rgb="rgb", depth="depth", acc="acc", z_vals="z_vals", weights="weights"), 2e-5)
    L, Lv = 7, 3
    gc = W.pdrf_grid_size(AABB[:3], AABB[3:], 24 ** 3)
    gf = W.pdrf_grid_size(AABB[:3], AABB[3:], 48 ** 3)
    ic, icv = W.pe_dim(L), W.pe_dim(Lv)
    vc = O.Voxel(W.make_pdrf_state_dict(74, gc, input_ch=32 + ic, input_ch_views=icv, hidden_dim=64, geo_feat_dim=15), "", gc, AABB,
                 input_ch=32 + ic, input_ch_views=icv, hidden_dim=64, geo_feat_dim=15, rgb_act="relu")
    vf = O.Voxel(W.make_pdrf_state_dict(75, gf, input_ch=64 + ic, input_ch_views=icv, hidden_dim=256, geo_feat_dim=128), "", gf, AABB,
                 input_ch=64 + ic, input_ch_views=icv, hidden_dim=256, geo_feat_dim=128, rgb_act="none")
    res = O.render_c2f(vc, vf, O.make_cfg(N_samples=64, N_importance=32, multires=L, multires_views=Lv), W.synthetic_rays(43, 48))
    for k in ("rgb0", "depth0", "acc0", "z_std", "z_vals0", "weights0"):
        assert maxabs(res[k], g["c_" + k]) < 5e-5, k
    same = np.abs(res["z_vals"] - g["c_z_vals"]).max(-1) < 5e-5           # rays whose importance samples agree (G9's rule)
    assert same.mean() > 0.8
    for k in ("rgb", "depth", "acc", "weights"):
        assert maxabs(res[k][same], g["c_" + k][same]) < 5e-5, k


def test_G25_pbe_composite_feature():
    """kernel_type='PBE' (renderer.py:30-34): the coarse level composites its geo features and runs its colour network per ray
    (voxnerf.py:223-239); eval render, coarse_render (renderer.py:468-592) and the level on explicit inputs vs the reference"""
    g = load_golden("G25_pbe_composite_feature")
    gc = W.pdrf_grid_size(AABB[:3], AABB[3:], 24 ** 3)
    gf = W.pdrf_grid_size(AABB[:3], AABB[3:], 48 ** 3)
    vc = O.Voxel(W.make_pdrf_state_dict(91, gc, input_ch=95, hidden_dim=64, geo_feat_dim=15), "", gc, AABB, input_ch=95, hidden_dim=64,
                 geo_feat_dim=15, rgb_act="relu", composite_feature=True)
    vf = O.Voxel(W.make_pdrf_state_dict(92, gf, input_ch=127, hidden_dim=256, geo_feat_dim=128), "", gf, AABB, input_ch=127, hidden_dim=256,
                 geo_feat_dim=128, rgb_act="none")
    lv = O.voxel_forward(vc, g["l_pts"], g["l_vd"], g["l_fts"], g["l_z"], g["l_rd"])
    for k in ("color", "depth", "acc", "weights", "feature"):
        assert maxabs(lv[k], g["l_" + k]) < 2e-5, k
    assert lv["feature"].shape == (24, 15)
    rays = W.synthetic_rays(51, 56)
    res = O.render_c2f(vc, None, O.make_cfg(N_samples=64, N_importance=0), rays, want_feature=True)
    assert maxabs(res["rgb"], g["coarse_rgb"]) < 2e-5
    assert maxabs(res["feature"].reshape(-1)[:56 * 15].reshape(56, 15), g["coarse_feat"]) < 2e-5
    res = O.render_c2f(vc, vf, O.make_cfg(N_samples=64, N_importance=32), rays)
    for k in ("rgb0", "depth0", "acc0", "z_std", "z_vals0", "weights0"):
        assert maxabs(res[k], g[k]) < 5e-5, k
    same = np.abs(res["z_vals"] - g["z_vals"]).max(-1) < 5e-5
    assert same.mean() > 0.8
    for k in ("rgb", "depth", "acc", "weights"):
        assert maxabs(res[k][same], g[k][same]) < 5e-5, k


def test_G10_rbk_weighted_sum():
    g = load_golden("G10_rbk_weighted_sum")
    ccw = g["ccw"]
    assert maxabs(O.weighted_sum(g["rgb"], ccw), g["o_rgb"]) < 1e-6
    assert maxabs(O.weighted_sum(g["depth"], ccw).reshape(-1), g["o_depth"]) < 1e-6
    assert maxabs(O.weighted_sum(g["acc"], ccw).reshape(-1), g["o_acc"]) < 1e-6
    for k in ("rgb0", "z_std", "weights", "depth_feature"):
        out = O.weighted_sum(g["ex_" + k], ccw)
        assert maxabs(out.reshape(g["o_" + k].shape), g["o_" + k]) < 1e-6, k


def test_G11_crf():
    g = load_golden("G11_crf")
    x, f2, f32 = g["x"], g["f2"], g["f32"]
    gam = O.Crf("gamma")
    ev = O.Crf("learn", W.make_crf_state_dict(41, 2), 2)
    assert maxabs(O.crf_forward(gam, x), g["rgb_gamma"]) < 1e-6
    assert maxabs(O.luma(O.crf_forward(ev, x, f2)), g["luma_learn_f2"]) < 1e-6
    assert maxabs(O.luma(O.crf_forward(ev, x, None)), g["luma_learn_nofeat"]) < 1e-6
    assert maxabs(O.luma(O.crf_forward(ev, x, f2, skip_learn=True)), g["luma_learn_skip"]) < 1e-6
    assert maxabs(O.crf_forward(ev, x, f32), g["tone_learn_f32"]) < 1e-6
    assert maxabs(np.repeat(O.luma(O.crf_forward(ev, x, f2)), 3, -1), g["luma_learn_keep"]) < 1e-6
    assert maxabs(O.luma(O.crf_forward(ev, x, None)), g["luma_chunked"]) < 1e-6
    ev0 = O.Crf("learn", W.make_crf_state_dict(43, 0), 0)
    assert maxabs(O.crf_forward(O.Crf("none"), x), g["rgb_none"]) == 0.0
    assert maxabs(O.luma(O.crf_forward(ev0, x)), g["luma_learn0"]) < 1e-6
    assert maxabs(O.luma(O.crf_forward(gam, x)), g["luma_gamma"]) < 1e-6
    assert maxabs(O.luma(O.crf_forward(gam, x), "rec709"), g["luma_gamma_rec709"]) < 1e-6
    assert maxabs(O.luma(O.crf_forward(gam, x), "avg"), g["luma_gamma_avg"]) < 1e-6


def test_G12_egm_loss():
    g = load_golden("G12_egm_loss")
    assert abs(O.egm_loss(g["ls"], g["le"], g["bii"]) - float(g["loss_plain"])) < 1e-5 * float(g["loss_plain"])
    v = O.egm_loss(g["ls3"], g["le3"], g["bii"], color_mask=g["cmask"])
    assert abs(v - float(g["loss_mask"])) < 1e-5 * float(g["loss_mask"])
    v = O.egm_loss(g["ls3"], g["le3"], g["bii"], color_mask=g["cmask"], color_weight=[0.4, 0.2, 0.4])
    assert abs(v - float(g["loss_mask_w"])) < 1e-5 * float(g["loss_mask_w"])


def test_G13_edi():
    g = load_golden("G13_edi")
    for s in range(8):
        img = O.bii_image(g[f"x{s}"], g[f"y{s}"], g[f"p{s}"], 16, 16, 0.2, 0.25, True)
        assert maxabs(img, g["bii"][s]) < 1e-6
        img = O.bii_image(np.floor(g[f"x{s}"]), np.floor(g[f"y{s}"]), g[f"p{s}"], 16, 16, 0.2, 0.25, False)
        assert maxabs(img, g[f"bii_ni{s}"]) < 1e-6
    assert maxabs(O.inner_double_integral(g["bii"]), g["inner"]) < 1e-6
    assert maxabs(O.deblur_double_integral(g["blurry"], g["bii"]), g["sharp"]) < 2e-6
    assert maxabs(O.deblur_double_integral(g["blurry3"], g["bii3"]), g["sharp3"]) < 2e-6


@pytest.mark.parametrize("cfg", ["blender", "cdavis"])
def test_G14_loss_assembly(cfg):
    g = load_golden("G14_loss_assembly")
    R, P = 64, 10
    flw, w_pts0, w_egm = [float(v) for v in g[f"{cfg}_scalars"]]
    rgb_p, rgb0_p, ccw = g[f"{cfg}_rgb_p"], g[f"{cfg}_rgb0_p"], g[f"{cfg}_ccw"]
    tgt, tgt0 = g[f"{cfg}_target"], g[f"{cfg}_target_pts0"]
    crf_rgb = O.Crf("gamma" if cfg == "blender" else "none")
    crf_ev = O.Crf("learn", W.make_crf_state_dict(51, 2), 2)
    enc = lambda x: O.crf_forward(crf_rgb, x)
    rgb, rgb1, awp = O.weighted_sum(rgb_p, ccw[0]), O.weighted_sum(rgb0_p, ccw[0]), O.weighted_sum(rgb_p, ccw[1])
    loss = O.mse(enc(rgb), tgt) + O.mse(enc(rgb1), tgt)
    loss = loss * (1 - flw) + O.mse(enc(awp), tgt) * flw
    pts0 = O.mse(enc(rgb_p.reshape(R, P, 3)[:, 0]), tgt0) + O.mse(enc(rgb0_p.reshape(R, P, 3)[:, 0]), tgt0)
    loss = loss + pts0 * w_pts0
    assert abs(loss - float(g[f"{cfg}_img_loss"])) < 2e-6
    assert abs(pts0 - float(g[f"{cfg}_pts0"])) < 2e-6
    cn, cp, cmask = g[f"{cfg}_cn"], g[f"{cfg}_cp"], g[f"{cfg}_cmask"]
    thr = 0.2 if cfg == "blender" else 0.25
    bii = (np.float32(thr) * cn + np.float32(thr) * cp).astype(np.float32)
    if cfg == "blender":
        feat = np.stack([cn, cp], -1)
        lum = lambda x: O.luma(O.crf_forward(crf_ev, x, feat))
        kw = {}
    else:
        fn = np.zeros((cn.shape[0], 3), np.float32)
        fp = np.zeros((cn.shape[0], 3), np.float32)
        fn[cmask] = cn
        fp[cmask] = cp
        feat = np.stack([fn, fp], -1)
        lum = lambda x: O.crf_forward(crf_ev, x, feat)
        kw = dict(color_mask=cmask, color_weight=[0.4, 0.2, 0.4])
    egm = O.egm_loss(lum(g[f"{cfg}_es0"]), lum(g[f"{cfg}_ee0"]), bii, **kw) + \
        O.egm_loss(lum(g[f"{cfg}_es"]), lum(g[f"{cfg}_ee"]), bii, **kw)
    assert abs(egm - float(g[f"{cfg}_egm"])) < 1e-5 * max(1.0, float(g[f"{cfg}_egm"]))
    total = loss + egm * w_egm
    assert abs(total - float(g[f"{cfg}_total"])) < 1e-5 * max(1.0, float(g[f"{cfg}_total"]))


def test_G15_awp_feature_integration():
    """G15: AdaptiveWeightProposal.feature_integration of the reference (awp.py:49-77)."""
    g = load_golden("G15_awp_feature_integration")
    for tag in ("a", "b"):
        feat = g[f"{tag}_feat"]
        out = O.awp_feature_integration(feat.reshape(-1, feat.shape[-2], feat.shape[-1]), g[f"{tag}_z"], g[f"{tag}_rays_d"])
        assert maxabs(out.reshape(g[f"{tag}_out"].shape), g[f"{tag}_out"]) < 2e-5 * max(1.0, np.abs(g[f"{tag}_out"]).max())



def test_G21_awp_sample_embed():
    """G21: AdaptiveWeightProposal.forward of the reference up to the input of motion_feature_embed_layer (awp.py:98-105): the oracle's
    sample embedding (evo_awp_sample_embed) and scan against the module's own intermediates; the float64 torch restatement the GPU
    tests differentiate (tests/test_gpu_awp.py _mlp64 / _scan64 restate the same lines) against the reference's autograd gradients."""
    import torch
    g = load_golden("G21_awp_sample_embed")
    sd = W.make_awp_embed_state_dict(211)
    ws = [sd[f"sample_feature_embed_layer.{l}.weight"] for l in range(4)]
    bs = [sd[f"sample_feature_embed_layer.{l}.bias"] for l in range(4)]
    x = g["depth_feature"]
    N, S, _ = x.shape
    h_local = O.awp_sample_embed(x, ws, bs).reshape(N, S, 64)
    assert maxabs(h_local, g["h_local"]) < 2e-5 * max(1.0, np.abs(g["h_local"]).max())
    h = O.awp_feature_integration(h_local, g["z"], g["rays_d"])
    assert maxabs(h.reshape(g["h"][..., :64].shape), g["h"][..., :64]) < 2e-5 * max(1.0, np.abs(g["h"]).max())
    # gradients: float64 restatement of awp.py:98-102 under torch.autograd vs the reference module's
    w64 = [torch.tensor(w, dtype=torch.float64, requires_grad=True) for w in ws]
    b64 = [torch.tensor(b, dtype=torch.float64, requires_grad=True) for b in bs]
    x64 = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    hh = x64
    for l in range(4):
        hh = torch.relu(hh @ w64[l].t() + b64[l])
    z, d = torch.tensor(g["z"], dtype=torch.float64), torch.tensor(g["rays_d"], dtype=torch.float64)
    dists = (z[..., 1:] - z[..., :-1]) * torch.norm(d[..., None, :], dim=-1)
    alpha = -torch.exp(-hh[..., :-1, :] * dists[..., None]) + 1
    alpha = torch.cat([alpha, torch.zeros_like(alpha[:, 0:1])], dim=-2)
    wts = alpha * torch.cumprod(torch.cat([torch.ones((N, 1, 64), dtype=torch.float64), -alpha + (1. + 1e-10)], -2), -1)[:, :-1, :]
    hint = torch.sum(wts * hh, dim=-2).reshape(g["proj"].shape)
    loss = (hint * torch.tensor(g["proj"], dtype=torch.float64)).sum()
    assert abs(loss.item() - float(g["loss"])) < 1e-4 * max(1.0, abs(float(g["loss"])))
    loss.backward()

    def rel(a, b):
        return float(np.linalg.norm(np.asarray(a, np.float64) - b) / max(np.linalg.norm(b), 1e-30))
    assert rel(x64.grad.numpy(), g["g.depth_feature"]) < 2e-4
    for l in range(4):
        assert rel(w64[l].grad.numpy(), g[f"g.w{l}"]) < 2e-4 and rel(b64[l].grad.numpy(), g[f"g.b{l}"]) < 2e-4

def test_G16_rbk_warp():
    """RigidBlurringModel.rbk_warp of the reference (blurmodel.py:51-82, rigid_warping.py)."""
    g = load_golden("G16_rbk_warp")
    for tag, M, uo in (("a", 9, True), ("b", 4, False), ("c", 9, True)):
        new_rays, tf = O.rbk_warp(g[f"{tag}_rays"], g[f"{tag}_r"], g[f"{tag}_v"], M, uo, want_transform=True)
        assert maxabs(new_rays, g[f"{tag}_new_rays"]) < 5e-6, tag
        assert maxabs(tf, g[f"{tag}_transform"]) < 5e-6, tag


def test_G17_compute_successor():
    """utils/events.py:72-120: integer outputs, bit-exact."""
    g = load_golden("G17_compute_successor")
    for tag in ("a", "b", "c"):
        succ, nsucc, latest, first = O.compute_successor(g[f"{tag}_ids"], g[f"{tag}_latest"].shape[0])
        assert np.array_equal(succ, g[f"{tag}_succ"]) and np.array_equal(nsucc, g[f"{tag}_nsucc"])
        assert np.array_equal(latest, g[f"{tag}_latest"]) and np.array_equal(first, g[f"{tag}_first"])


def _check_sample_events(fn, g, exact_rays):
    """fn(tag, hops) -> dict of numpy arrays; against golden G26 (both branches of sample_events)"""
    K = W.synthetic_camera()
    for tag in ("a", "b"):
        for branch, hops in (("1", None), ("2", g[f"{tag}_hops"])):
            out = fn(tag, hops, K)
            assert np.array_equal(out["events_pos_pol_cumsum"], g[f"{tag}_pos{branch}"]) and np.array_equal(out["events_neg_pol_cumsum"], g[f"{tag}_neg{branch}"]), (tag, branch)
            assert np.array_equal(out["events_coords_ids"], g[f"{tag}_cid"])
            for key, ref in (("events_rays_start", g[f"{tag}_rays_start"]), ("events_rays_end", g[f"{tag}_rays_end{branch}"])):
                assert out[key].shape == ref.shape == (g[f"{tag}_ids"].shape[0], 3, 2)
                if exact_rays:
                    assert np.array_equal(out[key], ref), (tag, branch, key)
                else:
                    assert np.abs(out[key] - ref).max() < 1e-6, (tag, branch, key)
            if branch == "2" and "successor" in out:
                assert np.array_equal(out["successor"], g[f"{tag}_succ2"])
            if f"{tag}_cmap" in g:
                assert np.array_equal(out["events_color_map"], g[f"{tag}_cmap_out"])


def test_G26_sample_events():
    """data/loader_events.py:259-304 on tables: polarity sums, ids and successors bit-exact; rays to float32 rounding of a 3-term sum"""
    g = load_golden("G26_sample_events")
    _check_sample_events(lambda tag, hops, K: O.sample_events(g[f"{tag}_events"], g[f"{tag}_coords"], g[f"{tag}_poses"], g[f"{tag}_ids"], K, hops=hops,
                                                             id_to_color_map=g[f"{tag}_cmap"] if f"{tag}_cmap" in g else None,
                                                             add_halfpix=bool(g[f"{tag}_halfpix"])), g, exact_rays=False)


def check_image_batch(out, g, exact_rays):
    """out: dict of numpy arrays with the reference's keys; against golden G28 (LLFFDataset.__getitem__, data/loader.py:325-356)"""
    for k in ("images_idx", "rgbsf", "rgbsf_pts0", "poses", "rays_x", "rays_y"):             # gathers and integer work: bit-exact
        ref = g["out_" + k]
        assert out[k].shape == ref.shape and out[k].dtype == ref.dtype, (k, out[k].shape, ref.shape, out[k].dtype, ref.dtype)
        assert np.array_equal(out[k], ref), k
    assert out["rays"].shape == g["out_rays"].shape
    assert np.array_equal(out["rays"][..., 0], g["out_rays"][..., 0])                      # origins: a gather
    err = np.abs(out["rays"][..., 1] - g["out_rays"][..., 1]).max()
    assert (err == 0) if exact_rays else (err < 1e-6), err


def test_G28_image_batch():
    g = load_golden("G28_image_batch")
    out = O.image_batch(g["ids"], g["images"], g["poses"], g["K"], pts0_images=g["pts0"])
    assert out["n_invalid"] == 0
    check_image_batch(out, g, exact_rays=False)
    assert "rgbsf_pts0" not in O.image_batch(g["ids"], g["images"], g["poses"], g["K"])
    bad = O.image_batch(np.array([-1, 5, g["images"][..., 0].size]), g["images"], g["poses"], g["K"])
    assert bad["n_invalid"] == 2 and list(bad["images_idx"][:, 0]) == [-1, 0, -1]


def test_G28_rays_without_the_half_pixel():
    """utils/rays.py:8-36 with add_halfpix=False (rectified event coordinates, data/loader_events.py:290-293)"""
    g = load_golden("G28_image_batch")
    o, d = O.get_rays_pix(g["nohalf_coords"], g["K"], g["nohalf_c2ws"], add_halfpix=False)
    assert np.array_equal(o, g["nohalf_pix_o"]) and np.abs(d - g["nohalf_pix_d"]).max() < 1e-6
    H, Wd = g["images"].shape[1:3]
    o, d = O.get_rays(H, Wd, g["K"], g["poses"][1], add_halfpix=False)
    assert np.array_equal(o, g["nohalf_full_o"]) and np.abs(d - g["nohalf_full_d"]).max() < 1e-6
    o1, d1 = O.get_rays_pix(g["nohalf_coords"], g["K"], g["nohalf_c2ws"])                  # the default differs
    assert np.abs(d1 - g["nohalf_pix_d"]).max() > 1e-3


POSE_TOL = 5e-7     # one float32 ulp of a pose entry in [2, 4): the float64 evaluations agree to ~1e-15, so at most the final rounding differs (the C oracle: 0 on both tracks)


def check_pose_track(interp, sample, g):
    """interp(tag, t) -> [n, 4, 4]; sample(tag) -> sample_events dict; against golden G29 (the reference's interpolate_poses and
    sample_events run on scipy)"""
    worst = 0.0
    for tag in ("a", "b"):
        got = interp(tag, g[f"{tag}_tq"])
        ref = g[f"{tag}_poses"]
        assert got.shape == ref.shape and got.dtype == np.float32
        assert np.array_equal(got[:, 3], ref[:, 3])
        err = np.abs(got - ref).max()
        worst = max(worst, err)
        assert err < POSE_TOL, (tag, err)
        out = sample(tag)
        assert np.array_equal(out["events_pos_pol_cumsum"], g[f"{tag}_pos"]) and np.array_equal(out["events_neg_pol_cumsum"], g[f"{tag}_neg"])
        assert np.array_equal(out["events_coords_ids"], g[f"{tag}_cid"])
        for key in ("events_rays_start", "events_rays_end"):
            ref = g[key.replace("events_", f"{tag}_")]
            assert out[key].shape == ref.shape
            assert np.abs(out[key] - ref).max() < 1e-5, (tag, key, np.abs(out[key] - ref).max())     # the ray direction scales a pose entry by up to ~1.2
    return worst


def test_G29_pose_track():
    g = load_golden("G29_pose_track")
    rc = lambda tag: g[f"{tag}_recenter_c2w"] if f"{tag}_recenter_c2w" in g else None
    K = W.synthetic_camera()
    check_pose_track(lambda tag, t: O.interpolate_poses(g[f"{tag}_key_t"], g[f"{tag}_key_poses"], t, float(g[f"{tag}_bd_scale"]), rc(tag)),
                     lambda tag: O.sample_events_track(g[f"{tag}_events"], g[f"{tag}_coords"], g[f"{tag}_key_t"], g[f"{tag}_key_poses"], g[f"{tag}_ids"], K,
                                                       bd_scale=float(g[f"{tag}_bd_scale"]), recenter_c2w=rc(tag), add_halfpix=bool(g[f"{tag}_intc"])), g)
    with pytest.raises(ValueError):
        O.interpolate_poses(g["a_key_t"][:3], g["a_key_poses"][:3], g["a_tq"][:4])


def test_pose_track_host_tables_against_the_oracle():
    """The product's host-side preparation (evdeblurnerf_amd/poses.py: quaternions, rotation vectors, spline polynomials -- numpy,
    no GPU) evaluated in numpy equals the oracle's from-raw-keys evaluation: the two restatements are independent code."""
    from evdeblurnerf_amd import poses as P
    g = load_golden("G29_pose_track")
    for tag in ("a", "b"):
        kt, kp = g[f"{tag}_key_t"], g[f"{tag}_key_poses"]
        q = P.quat_from_matrix(P._orthogonalize(kp[:, :, :3]))
        rv = P._as_rotvec(P._qmul(q[:-1] * np.array([-1.0, -1, -1, 1]), q[1:]))
        cf = P.notaknot_cubic(kt, kp[:, :, 3])
        t = np.clip(g[f"{tag}_tq"], kt[0], kt[-1])
        ind = np.clip(np.searchsorted(kt, t) - 1, 0, kt.shape[0] - 2)
        u = (t - kt[ind]) / np.diff(kt)[ind]
        T = cf[ind, 0] + u[:, None] * (cf[ind, 1] + u[:, None] * (cf[ind, 2] + u[:, None] * cf[ind, 3]))
        ref = O.interpolate_poses(kt, kp, g[f"{tag}_tq"], 1.0, None)
        assert np.abs(T.astype(np.float32) - ref[:, :3, 3]).max() < 1e-6
        ang = np.linalg.norm(rv[ind] * u[:, None], axis=-1)
        dq = np.concatenate([rv[ind] * u[:, None] * (np.sin(ang / 2) / np.maximum(ang, 1e-300))[:, None], np.cos(ang / 2)[:, None]], -1)
        qq = P._qmul(q[ind], dq)
        r00 = qq[:, 0] ** 2 - qq[:, 1] ** 2 - qq[:, 2] ** 2 + qq[:, 3] ** 2                  # R[0, 0] -> column 1 negated in the LLFF layout
        assert np.abs((-r00).astype(np.float32) - ref[:, 0, 1]).max() < 1e-6


# ---- gradients: the float64 torch restatements the GPU training tests compare against (tests/torch_restatement.py) are pinned
# here to gradients computed by torch.autograd ON THE REFERENCE MODULES (tools/gen_golden.py G18, G19)
def _check_grad(got, g, key, idx):
    from torch_restatement import grad_summary
    sm, head = grad_summary(got, 7000 + idx)
    ref_sm, ref_head = g[f"g.{key}.summary"], g[f"g.{key}.head"]
    norm = max(float(ref_sm[0]), 1e-30)
    assert abs(sm[0] - ref_sm[0]) < 2e-4 * norm, (key, sm[0], ref_sm[0])
    assert abs(sm[1] - ref_sm[1]) < 1e-3 * norm, (key, sm[1], ref_sm[1])                     # random projection: an O(norm) number
    assert np.abs(head - ref_head).max() < 2e-4 * max(np.abs(ref_head).max(), norm / np.sqrt(max(np.asarray(got).size, 1))), key
    if f"g.{key}.elem_idx" in g:             # element level: the 256 largest elements and 256 seeded random ones of the reference's gradient
        from torch_restatement import check_grad_elements
        err, ok = check_grad_elements(got, g[f"g.{key}.elem_idx"], g[f"g.{key}.elem_val"], 2e-4)
        assert ok, (key, err)


def test_G18_nerf_gradients_of_the_restatement_match_the_reference():
    import torch
    from torch_restatement import TorchNerf, nerf_composite
    g = load_golden("G18_nerf_grads")
    sd = W.make_nerf_state_dict(18, D=8, W=64, rgb_add_bias=True)
    o = torch.tensor(g["o"], dtype=torch.float64, requires_grad=True)
    d = torch.tensor(g["d"], dtype=torch.float64, requires_grad=True)
    z = torch.tensor(g["z"], dtype=torch.float64)
    R, S = z.shape
    pts = (o[:, None] + d[:, None] * z[..., None]).reshape(-1, 3)
    vd = d / d.norm(dim=-1, keepdim=True)
    net = TorchNerf(sd)
    raw = net(pts, vd[:, None].expand(-1, S, -1).reshape(-1, 3)).reshape(R, S, 4)
    rgb_map, _ = nerf_composite(raw, z, d)
    assert maxabs(rgb_map.detach().numpy(), g["rgb_map"]) < 2e-6
    (rgb_map * torch.tensor(g["w_rgb"], dtype=torch.float64)).sum().backward()
    keys = [k[2:-8] for k in g if k.startswith("g.") and k.endswith(".summary")]
    assert len(keys) == len(sd) + 2
    for idx, key in enumerate(keys):                                   # the generator's order = named_parameters() then the rays
        if key == "rays_o":
            got = o.grad
        elif key == "rays_d":
            got = d.grad
        else:
            got = net.p[key.replace(".", "_")].grad
        _check_grad(got.numpy(), g, key, idx)


def test_G19_c2f_gradients_of_the_restatement_match_the_reference():
    """the whole mode='c2f' training forward of the reference (NDC ray packing, both PDRF levels, resampled positions, TV) vs the
    restated pipeline on the reference's sample positions: every parameter gradient of both levels and the ray gradient"""
    import torch
    from evdeblurnerf_amd.renderer import NeRFAll
    from torch_restatement import TorchVoxLevel, c2f_pipeline, torch_tv
    g = load_golden("G19_c2f_grads")
    gc, gf = [int(v) for v in g["grid_coarse"]], [int(v) for v in g["grid_fine"]]
    sd = dict(W.prefixed(W.make_pdrf_state_dict(91, gc, input_ch=95, hidden_dim=64, geo_feat_dim=15, add_bias_color=True), "mlp_coarse"))
    sd.update(W.prefixed(W.make_pdrf_state_dict(92, gf, input_ch=127, hidden_dim=256, geo_feat_dim=128, add_bias_color=True), "mlp_fine"))
    aabb = ([-1.5, -1.5, -1.0], [1.5, 1.5, 1.0])
    rays = torch.tensor(g["rays"], dtype=torch.float64, requires_grad=True)
    rb = NeRFAll.ray_batch_train(400, 400, W.synthetic_camera(), rays)
    levels, grids = {}, {}
    for name in ("coarse", "fine"):
        pre = f"mlp_{name}."
        lsd = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
        levels[name] = TorchVoxLevel(lsd)
        grids[name] = ([torch.tensor(np.asarray(lsd[f"app_plane.{i}"]), dtype=torch.float64, requires_grad=True) for i in range(3)],
                       [torch.tensor(np.asarray(lsd[f"app_line.{i}"]), dtype=torch.float64, requires_grad=True) for i in range(3)],
                       torch.tensor(np.asarray(lsd["basis_mat.weight"]), dtype=torch.float64, requires_grad=True))
    z0, zm = torch.tensor(g["z_vals0"], dtype=torch.float64), torch.tensor(g["z_vals"], dtype=torch.float64)
    rgb0, rgb = c2f_pipeline(levels, grids, rb, z0, zm, aabb)
    assert maxabs(rgb.detach().numpy(), g["rgb"]) < 5e-6 and maxabs(rgb0.detach().numpy(), g["rgb0"]) < 5e-6
    tv = 5 * sum(torch_tv(grids[n_][0][i]) * 1e-2 + torch_tv(grids[n_][1][i]) * 1e-3 for n_ in ("coarse", "fine") for i in range(3))
    assert abs(tv.item() - float(g["tv"])) < 1e-5 * float(g["tv"])
    loss = (rgb * torch.tensor(g["w_rgb"], dtype=torch.float64)).sum() + (rgb0 * torch.tensor(g["w_rgb0"], dtype=torch.float64)).sum() + 0.1 * tv
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * max(1.0, abs(float(g["loss"])))
    loss.backward()
    keys = [k[2:-8] for k in g if k.startswith("g.") and k.endswith(".summary")]
    assert len(keys) == 2 * 15 + 1
    for idx, key in enumerate(keys):
        if key == "rays":
            got = rays.grad
        else:
            lvl, rest = key.split(".", 1)
            name = lvl[len("mlp_"):]
            if rest.startswith("app_plane."):
                got = grids[name][0][int(rest[-1])].grad
            elif rest.startswith("app_line."):
                got = grids[name][1][int(rest[-1])].grad
            elif rest == "basis_mat.weight":
                got = grids[name][2].grad
            else:
                got = levels[name].p[rest.replace(".", "_")].grad
        _check_grad(got.numpy(), g, key, idx)


def test_G22_mam():
    """G22: the reference's MotionAggregationModule run in training mode.  (a) the oracle's per-sample part (evo_mam_local: mam.py:72-74,
    29-33 as written) against the inputs the real module handed to Corr.conva / Corr.convb; (b) tools/awp_standin.py MAMLike -- the
    plain-torch module the GPU tests compare FusedAWP with, where the reference cannot be imported -- loads the reference's state dict
    and reproduces its output and its autograd gradients."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from awp_standin import MAMLike
    g = load_golden("G22_mam")
    R, P = g["x_global"].shape[:2]
    inter, intra = O.mam_local(g["x_local"], g["sd.linear.weight"], g["sd.linear.bias"], g["sd.Corr.line_conv_att.weight"], P)
    assert maxabs(inter, g["inter"]) < 1e-5 * max(1.0, np.abs(g["inter"]).max())
    assert maxabs(intra, g["intra"]) < 1e-5 * max(1.0, np.abs(g["intra"]).max())
    mam = MAMLike(32, P - 1).train()
    mam.load_state_dict({k[3:]: torch.tensor(g[k]) for k in g if k.startswith("sd.")}, strict=True)
    xg = torch.tensor(g["x_global"], requires_grad=True)
    xl = torch.tensor(g["x_local"], requires_grad=True)
    out = mam(xg, xl)
    assert maxabs(out.detach().numpy(), g["out"]) < 2e-5
    ps = [mam.linear.weight, mam.linear.bias, mam.Corr.line_conv_att.weight]
    grads = torch.autograd.grad((out * torch.tensor(g["proj"])).sum(), [xg, xl] + ps)
    for got, key in zip(grads, ["g.x_global", "g.x_local", "g.linear.weight", "g.linear.bias", "g.line_conv_att.weight"]):
        if key == "g.linear.bias":      # analytically zero (a constant added to every curve is removed by the training-mode BatchNorm): noise
            assert np.abs(got.numpy()).max() < 1e-4 and np.abs(g[key]).max() < 1e-4
            continue
        assert np.linalg.norm(got.numpy() - g[key]) < 2e-5 * np.linalg.norm(g[key]), key


def _g27_standin(g):
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from awp_standin import RefLikeAWP
    P, VF = g["out"].shape[1], g["view_feature"].shape[1]
    awp = RefLikeAWP(P=P, view_ch=VF, mam="corr")
    sd = {k[3:]: torch.tensor(g[k]) for k in g if k.startswith("sd.")}
    missing, unexpected = awp.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("sample_feature_embed_layer") for k in missing), (missing, unexpected)
    return awp


def test_G27_awp_per_ray():
    """G27: the reference's AdaptiveWeightProposal.forward with the real MotionAggregationModule, training mode.  (a) the oracle chain
    evo_awp_feature_integration -> evo_mam_local -> evo_awp_per_ray on the h_local the real forward produced: the proposal weights, the
    BatchNorm running estimates after the step, the eval-mode output; (b) tools/awp_standin.py RefLikeAWP(mam="corr") -- what the GPU tests
    compare the kernels with at real sizes -- loads the reference's state dict and reproduces the output and every autograd gradient."""
    import torch
    g = load_golden("G27_awp_per_ray")
    R, P = g["out"].shape
    S = g["z"].shape[1]
    sd = {k[3:]: g[k] for k in g if k.startswith("sd.")}
    h = O.awp_feature_integration(g["h_local"], g["z"], g["rays_d"]).reshape(R, P, -1)
    d0 = g["rays_d"].reshape(R, P, 3)[:, 0]
    view = np.concatenate([g["view_feature"], O.embed(d0 / np.linalg.norm(d0, axis=-1, keepdims=True), 2)], -1)
    inter, intra = O.mam_local(g["h_local"], sd["MAM.linear.weight"], sd["MAM.linear.bias"], sd["MAM.Corr.line_conv_att.weight"], P)
    out, mean, var = O.awp_per_ray(h, view, inter, intra, sd, training=True)
    assert maxabs(out, g["out"]) < 2e-6, maxabs(out, g["out"])
    rm = 0.9 * sd["MAM.Corr.convd.1.running_mean"] + 0.1 * mean          # BatchNorm1d momentum 0.1 (torch default, mam.py:26)
    rv = 0.9 * sd["MAM.Corr.convd.1.running_var"] + 0.1 * var
    assert maxabs(rm, g["after.running_mean"]) < 1e-6 and maxabs(rv, g["after.running_var"]) < 1e-6
    sd2 = dict(sd)
    sd2["MAM.Corr.convd.1.running_mean"], sd2["MAM.Corr.convd.1.running_var"] = g["after.running_mean"], g["after.running_var"]
    out_e, _, _ = O.awp_per_ray(h, view, inter, intra, sd2, training=False)
    assert maxabs(out_e, g["out_eval"]) < 2e-6
    # (b)
    awp = _g27_standin(g).train()
    hl, rd, vf = (torch.tensor(g[k], requires_grad=True) for k in ("h_local", "rays_d", "view_feature"))
    o = awp.forward_from_local(hl, torch.tensor(g["z"]), rd, vf)
    assert maxabs(o.detach().numpy(), g["out"]) < 2e-6
    names = [k[2:] for k in g if k.startswith("g.") and k[2:] not in ("h_local", "rays_d", "view_feature")]
    pd = dict(awp.named_parameters())
    grads = torch.autograd.grad((o * torch.tensor(g["proj"])).sum(), [hl, rd, vf] + [pd[k] for k in names])
    for got, key in zip(grads, ["h_local", "rays_d", "view_feature"] + names):
        ref = g["g." + key]
        if key == "MAM.linear.bias":     # analytically zero (the training-mode BatchNorm removes a constant added to every curve)
            assert np.abs(got.numpy()).max() < 1e-4 and np.abs(ref).max() < 1e-4
            continue
        assert np.linalg.norm(got.numpy() - ref) < 3e-5 * np.linalg.norm(ref), (key, np.linalg.norm(got.numpy() - ref) / np.linalg.norm(ref))
    assert maxabs(awp.MAM.Corr.convd[1].running_mean.numpy(), g["after.running_mean"]) < 1e-6


def test_G31_event_hop_schedule_and_draw():
    """evdeblurnerf_amd.events.annealing_interpolator / draw_hops (host logic of EventSampler.sample_events(events_ids, global_step=...))
    against the reference's utils/misc.py functions as loader_events.py:259-268 composes them (golden G31; no GPU: the draw runs on
    the CPU generator)."""
    import torch
    from evdeblurnerf_amd.events import annealing_interpolator, draw_hops
    g = load_golden("G31_event_hops")
    for m in ("linear", "cosine", "constant"):
        got = np.array([[float(annealing_interpolator(int(a), int(b), int(e), m)(int(st))) for st in g["steps"]] for a, b, e in g["cases"]])
        assert np.array_equal(got, g[f"interp_{m}"]), m
    for i, (mn, mx) in enumerate(g["draws"]):
        torch.manual_seed(3100 + i)
        hops = draw_hops(torch.tensor(g[f"ns_{i}"]), int(mn), int(mx))
        assert hops.dtype == torch.int64 and np.array_equal(hops.numpy(), g[f"hops_{i}"]), (i, mn, mx)


def _train_call_voxels(seed, g):
    gc, gf = [int(v) for v in g["grid_coarse"]], [int(v) for v in g["grid_fine"]]
    assert gc == W.pdrf_grid_size(AABB[:3], AABB[3:], 24 ** 3) and gf == W.pdrf_grid_size(AABB[:3], AABB[3:], 48 ** 3)
    sd = W.make_train_call_state_dict(seed, gc, gf)
    vc = O.Voxel(sd, "mlp_coarse.", gc, AABB, input_ch=95, hidden_dim=64, geo_feat_dim=15, rgb_act="relu")
    vf = O.Voxel(sd, "mlp_fine.", gf, AABB, input_ch=127, hidden_dim=256, geo_feat_dim=128, rgb_act="none")
    return vc, vf, sd


def _train_call_awp_sd(seed, g):
    sd = {k[len("awp.sd."):]: g[k] for k in g if k.startswith("awp.sd.")}
    sd.update(W.make_awp_embed_state_dict(seed * 10 + 1))
    return sd


def test_G32_train_forward():
    """G32: the reference's NeRFAll.forward in training mode with its real RigidBlurringModel and AdaptiveWeightProposal
    (networks/renderer.py:277-392).  The oracle's composition (oracle.train_forward) on the kernel's recorded outputs reproduces every
    output of the call: rgb, rgb1, rgb_awp, both pts0 tensors, the TV term, and the BatchNorm estimates the AWP leaves behind.
    Tolerances: sample_pdf's (u - cdf) / denom amplifies the cdf's rounding by 1 / pdf (conftest.sample_pdf_flip_report), and with 16 + 16
    samples on dense fields an importance sample that moves by 4e-4 moves a colour by 7e-5: pixels whose P rays carry the golden's sample
    positions to 5e-6 hold 2e-5 on every output, all others 5e-4; the proposal weights (BatchNorm over ALL pixels) 3e-4."""
    g = load_golden("G32_train_forward")
    vc, vf, sd = _train_call_voxels(32, g)
    awp_sd = _train_call_awp_sd(32, g)
    cfg = O.make_cfg(N_samples=16, N_importance=16, is_train=True)
    out = O.train_forward(vc, vf, cfg, g["new_rays"], g["weight"], g["img_embed"], awp_sd, sd)
    tight, errs_tight, errs_all = train_call_errors(out, out["render"]["z_vals"], g)
    assert tight.sum() >= 8, tight.sum()
    assert max(errs_tight.values()) < 2e-5, errs_tight
    assert max(errs_all.values()) < 5e-4, errs_all
    assert maxabs(O.ray_batch(cfg, g["new_rays"].reshape(-1, 3, 2))[:, 3:6], g["awp_in_rays_d"]) < 2e-6     # the AWP sees the NDC directions (:464-465)
    assert abs(out["tv"] - float(g["tv"])) < 2e-6 * max(1.0, abs(float(g["tv"])))
    assert maxabs(g["stage1_img_embed"], g["img_embed"]) == 0.0               # other_tensors carries the kernel's extras under stage1_ (:368)
    ref_ccw = g["awp_out"] + g["awp_out"] * np.float32(0.05)
    assert maxabs(out["ccw_fine"], ref_ccw / ref_ccw.sum(-1, keepdims=True)) < 3e-4
    rm = 0.9 * awp_sd["MAM.Corr.convd.1.running_mean"] + 0.1 * out["bn_mean"]
    rv = 0.9 * awp_sd["MAM.Corr.convd.1.running_var"] + 0.1 * out["bn_var"]
    assert maxabs(rm, g["awp.after.running_mean"]) < 1e-4 and maxabs(rv, g["awp.after.running_var"]) < 1e-4
    # the projection the gradients were taken of (a second, independent check of the recorded loss)
    loss = sum(float((out[k].astype(np.float64) * g["proj." + k]).sum()) for k in errs_all) + 0.1 * out["tv"]
    assert abs(loss - float(g["loss"])) < 5e-3, (loss, float(g["loss"]))


def test_G35_llff_pose_preparation():
    """G35: LLFFDataset.load_poses + recenter_poses (data/loader.py:178-216, utils/data.py:115-183) -- host numpy in the reference and
    here (a few dozen 3 x 5 matrices): evdeblurnerf_amd.loader.load_poses / recenter_poses / poses_avg, bit-equal (same numpy calls)."""
    from evdeblurnerf_amd import loader as LD
    g = load_golden("G35_llff_poses")
    for tag in ("a", "b"):
        factor, bdf, H, Wd = g[f"{tag}_args"]
        poses, bds, sc = LD.load_poses(g[f"{tag}_poses_bounds"], int(factor), (int(H), int(Wd), 3), bd_factor=None if bdf < 0 else float(bdf))
        assert poses.dtype == np.float32 and np.array_equal(poses, g[f"{tag}_poses"]) and np.array_equal(bds, g[f"{tag}_bds"]) and float(sc) == float(g[f"{tag}_sc"])
        rec, c2w = LD.recenter_poses(poses, return_c2w=True)
        assert maxabs(c2w, g[f"{tag}_c2w"]) < 1e-12 and maxabs(rec, g[f"{tag}_recentered"]) < 1e-6
        assert maxabs(LD.recenter_poses(poses, c2w=g[f"{tag}_c2w"]), g[f"{tag}_recentered"]) < 1e-6


def check_event_tables(got, g, tag):
    """the tables of load_event_data, bit for bit (integer / index work)"""
    ev = np.asarray(got["events"], np.float64)
    assert ev.shape == g[f"{tag}_events"].shape and np.array_equal(ev, g[f"{tag}_events"])
    assert np.array_equal(np.asarray(got["id_to_coords"], np.float64), g[f"{tag}_id_to_coords"])
    assert np.array_equal(np.asarray(got["id_to_color_map"]).astype(np.uint8), g[f"{tag}_id_to_color_map"])
    assert np.array_equal(np.asarray(got["events_num_successors"]).astype(np.int64), g[f"{tag}_num_successors"])
    assert np.array_equal(np.asarray(got["events_with_successor_idx"]).astype(np.int64), g[f"{tag}_with_successor_idx"])
    assert bool(got["intcoords"]) == (tag == "int")


@pytest.mark.parametrize("tag", ["int", "flt"])
def test_G34_event_tables(tag):
    """G34: LLFFEventsDataset.load_event_data run by the generator on arrays (data/loader_events.py:150-257): the oracle's array-level
    restatement gives the same event table, coordinate ids in np.unique's byte order, Bayer colour map (integer pixels / ev_map inverse
    maps), successor counts and the start-event list."""
    g = load_golden("G34_event_tables")
    h, w = (int(v) for v in g[f"{tag}_hw"])
    acc = [int(v) for v in g[f"{tag}_acc"]]
    min_step = max(acc[0], acc[2]) if tuple(acc[:2]) != (0, 0) else 0
    ev_map = (g["flt_inv_mapx"], g["flt_inv_mapy"]) if tag == "flt" else None
    got = O.event_tables(g[f"{tag}_x"], g[f"{tag}_y"], g[f"{tag}_t"], g[f"{tag}_p"], h, w, float(g[f"{tag}_key_t"].min()), float(g[f"{tag}_key_t"].max()),
                         ev_map=ev_map, color_events=True, min_step=min_step)
    check_event_tables(got, g, tag)
    if tag == "flt":            # coords_to_id of :201 (a dict for float coordinates): the same (x, y) -> id pairs
        c2i = g["flt_coords_to_id"]
        assert np.array_equal(got["id_to_coords"][c2i[:, 2].astype(np.int64)], c2i[:, :2])
