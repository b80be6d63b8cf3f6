"""ctypes front-end of the CPU oracle (oracle/evd_oracle.c).

TEST INFRASTRUCTURE ONLY: import from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never from the evdeblurnerf_amd package. numpy in,
numpy out; every function is a thin call into the C restatement, whose
comments cite the reference file:line it follows.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "build", "libevd_oracle.so")
MAXL = 16

ACT = {"none": 0, "relu": 1, "sigmoid": 2, "exp": 3, "sigmoid1": 4, "softplus": 5, "tanh": 6}
_fp = C.POINTER(C.c_float)


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("evd_oracle.c", "evd_oracle.h", "Makefile")]
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src)
    if force or stale:
        subprocess.run(["make", "-C", _HERE] + (["-B"] if force else []), check=True, capture_output=True)
    return _SO


def use_fast_build() -> str:
    """Switch this module to the THROUGHPUT build of the same C file (make fast: -O3 -march=native, fused multiply-adds allowed),
    compiled for the host it runs on.  For bench.py's cpu_baseline leg only: the checker build above keeps mul / add unfused
    (-ffp-contract=off) so that the restatement rounds like torch's float32 ops; a baseline should not be handicapped by that."""
    global _lib
    subprocess.run(["make", "-C", _HERE, "-B", "fast"], check=True, capture_output=True)      # always rebuilt: -march=native of THIS host
    _lib = None
    globals()["_SO_ACTIVE"] = os.path.join(_HERE, "build", "libevd_oracle_fast.so")
    return globals()["_SO_ACTIVE"]


class NerfStruct(C.Structure):
    _fields_ = [("D", C.c_int), ("W", C.c_int), ("input_ch", C.c_int), ("input_ch_views", C.c_int),
                ("skip", C.c_int), ("use_viewdirs", C.c_int), ("output_ch", C.c_int),
                ("pts_w", _fp * MAXL), ("pts_b", _fp * MAXL),
                ("views_w", _fp), ("views_b", _fp), ("feature_w", _fp), ("feature_b", _fp),
                ("alpha_w", _fp), ("alpha_b", _fp), ("rgb_w", _fp), ("rgb_b", _fp),
                ("output_w", _fp), ("output_b", _fp),
                ("rgb_act", C.c_int), ("sigma_act", C.c_int), ("rmnear", C.c_float)]


class VoxelStruct(C.Structure):
    _fields_ = [("num_layers", C.c_int), ("hidden_dim", C.c_int), ("geo_feat_dim", C.c_int),
                ("num_layers_color", C.c_int), ("input_ch", C.c_int), ("input_ch_views", C.c_int),
                ("app_dim", C.c_int), ("n_comp", C.c_int * 3), ("grid", C.c_int * 3), ("app_act", C.c_int),
                ("rgb_act", C.c_int), ("sigma_act", C.c_int), ("composite_feature", C.c_int),
                ("aabb", C.c_float * 6),
                ("sigma_w", _fp * MAXL), ("color_w", _fp * MAXL), ("color_b", _fp * MAXL),
                ("plane", _fp * 3), ("line", _fp * 3), ("basis", _fp), ("rmnear", C.c_float)]


class CrfStruct(C.Structure):
    _fields_ = [("map_type", C.c_int), ("gamma", C.c_float), ("extra_features", C.c_int),
                ("w", _fp * 4), ("b", _fp * 4)]


class RenderCfg(C.Structure):
    _fields_ = [("H", C.c_int), ("W", C.c_int), ("focal", C.c_float),
                ("ndc", C.c_int), ("use_viewdirs", C.c_int), ("lindisp", C.c_int), ("N_samples", C.c_int),
                ("N_importance", C.c_int), ("white_bkgd", C.c_int), ("multires", C.c_int),
                ("multires_views", C.c_int), ("near", C.c_float), ("far", C.c_float), ("perturb", C.c_float),
                ("is_train", C.c_int)]


class RenderOut(C.Structure):
    _fields_ = [(k, _fp) for k in ("rgb", "depth", "acc", "z_vals", "weights", "rgb0", "depth0", "acc0", "z_std",
                                   "z_vals0", "weights0", "feature", "raw")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(globals().get("_SO_ACTIVE") or build())
        _lib.evo_mse.restype = C.c_double
        _lib.evo_egm_loss.restype = C.c_double
        _lib.evo_tv_loss.restype = C.c_double
        _lib.evo_num_threads.restype = C.c_int
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(_fp) if a is not None else None


class Nerf:
    """Holds an evo_nerf struct and keeps the numpy parameter arrays alive."""

    def __init__(self, sd, prefix="", D=8, W=256, input_ch=63, input_ch_views=27, skip=4, use_viewdirs=True,
                 rgb_act="sigmoid", sigma_act="relu", rmnear=0.0, output_ch=4):
        g = lambda k: _f(sd[prefix + k]) if (prefix + k) in sd else None
        self.keep = []
        s = NerfStruct()
        s.D, s.W, s.input_ch, s.input_ch_views, s.skip = D, W, input_ch, input_ch_views, skip
        s.use_viewdirs, s.output_ch = int(use_viewdirs), output_ch
        for i in range(D):
            w, b = g(f"pts_linears.{i}.weight"), g(f"pts_linears.{i}.bias")
            self.keep += [w, b]
            s.pts_w[i], s.pts_b[i] = _p(w), _p(b)
        for name in ("views", "feature", "alpha", "rgb", "output"):
            key = {"views": "views_linears.0"}.get(name, f"{name}_linear")
            w, b = g(key + ".weight"), g(key + ".bias")
            self.keep += [w, b]
            setattr(s, name + "_w", _p(w))
            setattr(s, name + "_b", _p(b))
        s.rgb_act, s.sigma_act, s.rmnear = ACT[rgb_act], ACT[sigma_act], float(rmnear)
        self.s = s


class Voxel:
    def __init__(self, sd, prefix, grid, aabb, input_ch, num_layers=2, hidden_dim=64, geo_feat_dim=15,
                 num_layers_color=3, input_ch_views=27, app_dim=32, n_comp=(64, 16, 16), app_act="none",
                 rgb_act="relu", sigma_act="relu", composite_feature=False, rmnear=0.0):
        g = lambda k: _f(sd[prefix + k]) if (prefix + k) in sd else None
        self.keep = []
        s = VoxelStruct()
        s.num_layers, s.hidden_dim, s.geo_feat_dim, s.num_layers_color = num_layers, hidden_dim, geo_feat_dim, num_layers_color
        s.input_ch, s.input_ch_views, s.app_dim = input_ch, input_ch_views, app_dim
        for i in range(3):
            s.n_comp[i], s.grid[i] = n_comp[i], int(grid[i])
        for i in range(6):
            s.aabb[i] = float(aabb[i])
        s.app_act, s.rgb_act, s.sigma_act = ACT[app_act], ACT[rgb_act], ACT[sigma_act]
        s.composite_feature, s.rmnear = int(composite_feature), float(rmnear)
        for l in range(num_layers):
            w = g(f"sigma_net.{l}.weight")
            self.keep.append(w)
            s.sigma_w[l] = _p(w)
        for l in range(num_layers_color):
            w, b = g(f"color_net.{l}.weight"), g(f"color_net.{l}.bias")
            self.keep += [w, b]
            s.color_w[l], s.color_b[l] = _p(w), _p(b)
        for i in range(3):
            pl, li = g(f"app_plane.{i}"), g(f"app_line.{i}")
            self.keep += [pl, li]
            s.plane[i], s.line[i] = _p(pl), _p(li)
        b = g("basis_mat.weight")
        self.keep.append(b)
        s.basis = _p(b)
        self.s = s


class Crf:
    def __init__(self, map_type, sd=None, extra_features=0, gamma=2.2):
        s = CrfStruct()
        s.map_type = {"none": 0, "gamma": 1, "learn": 2}[map_type]
        s.gamma, s.extra_features = gamma, extra_features
        self.keep = []
        if map_type == "learn":
            for j, idx in enumerate((0, 2, 4, 6)):
                w, b = _f(sd[f"linear.{idx}.weight"]), _f(sd[f"linear.{idx}.bias"])
                self.keep += [w, b]
                s.w[j], s.b[j] = _p(w), _p(b)
        self.s = s


def embed(x, L):
    x = _f(x)
    n, dim = x.shape
    out = np.empty((n, dim * (1 + 2 * L)), np.float32)
    lib().evo_embed(_p(x), C.c_long(n), dim, L, _p(out))
    return out


def nerf_mlp(net: Nerf, emb, want_after=False, want_before=False):
    emb = _f(emb)
    n = emb.shape[0]
    raw = np.empty((n, 4 if net.s.use_viewdirs else net.s.output_ch), np.float32)
    fa = np.empty((n, net.s.W), np.float32) if want_after else None
    fb = np.empty((n, net.s.W), np.float32) if want_before else None
    lib().evo_nerf_mlp(C.byref(net.s), _p(emb), C.c_long(n), _p(raw), _p(fa), _p(fb))
    return raw, fa, fb


def composite(raw, z, rays_d, sigma_ch=3, rgb_ch0=0, n_rgb=3, rgb_act="sigmoid", sigma_act="relu",
              white_bkgd=False, rmnear_thresh=0.0, noise=None, feature=None):
    raw, z, rays_d = _f(raw), _f(z), _f(rays_d)
    R, S, Cc = raw.shape
    out = np.empty((R, n_rgb), np.float32)
    dens = np.empty((R, S - 1), np.float32)
    acc = np.empty((R,), np.float32)
    wts = np.empty((R, S), np.float32)
    depth = np.empty((R,), np.float32)
    noise = _f(noise) if noise is not None else None
    F = 0
    fmap = None
    if feature is not None:
        feature = _f(feature)
        F = feature.shape[-1]
        fmap = np.empty((R, F), np.float32)
    lib().evo_composite(_p(raw), _p(z), _p(rays_d), C.c_long(R), S, Cc, sigma_ch, rgb_ch0, n_rgb,
                        ACT[rgb_act], ACT[sigma_act], int(white_bkgd), C.c_float(rmnear_thresh), _p(noise),
                        _p(out), _p(dens), _p(acc), _p(wts), _p(depth), _p(feature), F, _p(fmap))
    return dict(rgb=out, density=dens, acc=acc, weights=wts, depth=depth, fmap=fmap)


def sample_pdf(bins, w, N, det=True, u=None):
    bins, w = _f(bins), _f(w)
    R, nb = bins.shape
    out = np.empty((R, N), np.float32)
    u = _f(u) if u is not None else None
    lib().evo_sample_pdf(_p(bins), _p(w), C.c_long(R), nb, N, int(det), _p(u), _p(out))
    return out


def get_rays(H, W, K, c2w, add_halfpix=True):
    K, c2w = _f(K), _f(c2w)
    o = np.empty((H, W, 3), np.float32)
    d = np.empty((H, W, 3), np.float32)
    lib().evo_get_rays(H, W, _p(K), _p(c2w), int(bool(add_halfpix)), _p(o), _p(d))
    return o, d


def get_rays_pix(coords, K, c2ws, add_halfpix=True):
    coords, K, c2ws = _f(coords), _f(K), _f(c2ws)
    n = coords.shape[0]
    o = np.empty((n, 3), np.float32)
    d = np.empty((n, 3), np.float32)
    lib().evo_get_rays_pix(_p(coords), _p(K), _p(c2ws), C.c_long(n), int(bool(add_halfpix)), _p(o), _p(d))
    return o, d


def ndc_rays(H, W, focal, near, o, d):
    o, d = _f(o), _f(d)
    n = o.shape[0]
    oo = np.empty_like(o)
    od = np.empty_like(d)
    lib().evo_ndc_rays(H, W, C.c_float(focal), C.c_float(near), _p(o), _p(d), C.c_long(n), _p(oo), _p(od))
    return oo, od


def make_cfg(H=400, W=400, focal=400.0, ndc=True, use_viewdirs=True, lindisp=False, N_samples=64, N_importance=0,
             white_bkgd=False, multires=10, multires_views=4, near=0.0, far=1.0, perturb=0.0, is_train=False):
    c = RenderCfg()
    c.H, c.W, c.focal = H, W, focal
    c.ndc, c.use_viewdirs, c.lindisp = int(ndc), int(use_viewdirs), int(lindisp)
    c.N_samples, c.N_importance, c.white_bkgd = N_samples, N_importance, int(white_bkgd)
    c.multires, c.multires_views = multires, multires_views
    c.near, c.far, c.perturb, c.is_train = near, far, perturb, int(is_train)
    return c


def ray_batch(cfg, rays):
    rays = _f(rays)
    R = rays.shape[0]
    rb = np.empty((R, 11), np.float32)
    nc = C.c_int(0)
    lib().evo_ray_batch(C.byref(cfg), _p(rays), C.c_long(R), _p(rb), C.byref(nc))
    return rb.reshape(-1)[: R * nc.value].reshape(R, nc.value).copy()


def _render(fn, coarse, fine, cfg, rays, t_rand, u, feat_dim, want_raw):
    rays = _f(rays)
    R = rays.shape[0]
    S, Ni = cfg.N_samples, cfg.N_importance
    St = S + Ni
    res = dict(rgb=np.empty((R, 3), np.float32), depth=np.empty((R,), np.float32), acc=np.empty((R,), np.float32),
               z_vals=np.empty((R, St), np.float32), weights=np.empty((R, St), np.float32))
    if Ni > 0:
        res.update(rgb0=np.empty((R, 3), np.float32), depth0=np.empty((R,), np.float32),
                   acc0=np.empty((R,), np.float32), z_std=np.empty((R,), np.float32),
                   z_vals0=np.empty((R, S), np.float32), weights0=np.empty((R, S), np.float32))
    if feat_dim:
        res["feature"] = np.empty((R, St, feat_dim), np.float32)
    if want_raw:
        res["raw"] = np.empty((R, St, 4), np.float32)
    out = RenderOut()
    for k, v in res.items():
        setattr(out, k, _p(v))
    t_rand = _f(t_rand) if t_rand is not None else None
    u = _f(u) if u is not None else None
    fn(C.byref(coarse.s), C.byref(fine.s) if fine is not None else None, C.byref(cfg), _p(rays), C.c_long(R),
       _p(t_rand), _p(u), C.byref(out))
    return res


def render_nerf(coarse: Nerf, fine, cfg, rays, t_rand=None, u=None, want_feature=False, want_raw=False):
    return _render(lib().evo_render_nerf, coarse, fine, cfg, rays, t_rand, u,
                   coarse.s.W if want_feature else 0, want_raw)


def render_c2f(coarse: Voxel, fine, cfg, rays, t_rand=None, u=None, want_feature=False):
    fd = (fine.s.geo_feat_dim if (fine is not None and cfg.N_importance > 0) else coarse.s.geo_feat_dim)
    return _render(lib().evo_render_c2f, coarse, fine, cfg, rays, t_rand, u, fd if want_feature else 0, False)


def voxel_forward(v: Voxel, pts, viewdirs, fts, z, rays_d, multires=10, multires_views=4, is_train=False):
    """VoxelNeRFBase.forward (voxnerf.py:210-259) on explicit inputs -> dict(color, depth, acc, weights, feature); the feature map is
    [R,S,geo], or the composited [R,geo] of a composite_feature level"""
    pts, viewdirs, fts, z, rays_d = _f(pts), _f(viewdirs), _f(fts), _f(z), _f(rays_d)
    R, S = z.shape
    G = v.s.geo_feat_dim
    res = dict(color=np.empty((R, 3), np.float32), depth=np.empty((R,), np.float32), acc=np.empty((R,), np.float32),
               weights=np.empty((R, S), np.float32), feature=np.empty((R, G) if v.s.composite_feature else (R, S, G), np.float32))
    lib().evo_voxel_forward(C.byref(v.s), _p(pts), _p(viewdirs), _p(fts), fts.shape[-1], _p(z), _p(rays_d), C.c_long(R), S, multires, multires_views,
                            int(bool(is_train)), _p(res["color"]), _p(res["depth"]), _p(res["acc"]), _p(res["weights"]), _p(res["feature"]))
    return res


def appfeature(v: Voxel, pts):
    pts = _f(pts).reshape(-1, 3)
    out = np.empty((pts.shape[0], v.s.app_dim), np.float32)
    lib().evo_appfeature(C.byref(v.s), _p(pts), C.c_long(pts.shape[0]), _p(out))
    return out


def weighted_sum(x, ccw):
    x, ccw = _f(x), _f(ccw)
    R, P = ccw.shape
    Cc = int(np.prod(x.shape[1:])) if x.ndim > 1 else 1
    out = np.empty((R, Cc), np.float32)
    lib().evo_weighted_sum(_p(x), _p(ccw), C.c_long(R), P, Cc, _p(out))
    return out.reshape((R,) + tuple(x.shape[1:]))


def compute_successor(ids, hw):
    """utils/events.py:72-120 on flat pixel ids -> (successor int64 [N], num_successors int32 [N], latest_seen, first_seen int64 [hw])"""
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    n = ids.shape[0]
    succ, nsucc = np.empty(n, np.int64), np.empty(n, np.int32)
    latest, first = np.empty(hw, np.int64), np.empty(hw, np.int64)
    ip, lp = C.POINTER(C.c_int), C.POINTER(C.c_longlong)
    lib().evo_compute_successor(ids.ctypes.data_as(ip), C.c_long(n), C.c_long(hw), succ.ctypes.data_as(lp), nsucc.ctypes.data_as(ip),
                                latest.ctypes.data_as(lp), first.ctypes.data_as(lp))
    return succ, nsucc, latest, first


def sample_events(events, id_to_coords, poses, events_ids, K, hops=None, id_to_color_map=None, add_halfpix=True):
    """data/loader_events.py:259-304 on tables: events float64 [N, ncol], id_to_coords [Nc, 2], poses [N, 3, 4], ids int64 [n]
    -> dict with the reference's keys (+ 'successor')"""
    ev = np.ascontiguousarray(events, dtype=np.float64)
    ids = np.ascontiguousarray(events_ids, dtype=np.int64)
    co, po, Kf = _f(id_to_coords), _f(poses), _f(np.asarray(K).reshape(-1))
    n, N, ncol = ids.shape[0], ev.shape[0], ev.shape[1]
    hp = np.ascontiguousarray(hops, dtype=np.int64) if hops is not None else None
    cm = np.ascontiguousarray(id_to_color_map, dtype=np.uint8) if id_to_color_map is not None else None
    rs, re = np.empty((n, 3, 2), np.float32), np.empty((n, 3, 2), np.float32)
    pos, neg = np.empty(n, np.float32), np.empty(n, np.float32)
    cid, succ = np.empty(n, np.int64), np.empty(n, np.int64)
    cmo = np.empty((n, 3), np.uint8) if cm is not None else None
    lp, bp, dp = C.POINTER(C.c_longlong), C.POINTER(C.c_ubyte), C.POINTER(C.c_double)
    lib().evo_sample_events(ev.ctypes.data_as(dp), C.c_long(N), ncol, _p(co), cm.ctypes.data_as(bp) if cm is not None else None, _p(po),
                            ids.ctypes.data_as(lp), hp.ctypes.data_as(lp) if hp is not None else None, C.c_long(n), _p(Kf), int(add_halfpix),
                            _p(rs), _p(re), _p(pos), _p(neg), cid.ctypes.data_as(lp), cmo.ctypes.data_as(bp) if cmo is not None else None,
                            succ.ctypes.data_as(lp))
    return {"events_pos_pol_cumsum": pos, "events_neg_pol_cumsum": neg, "events_rays_start": rs, "events_rays_end": re,
            "events_coords_ids": cid, "events_color_map": cmo.astype(bool) if cmo is not None else None, "successor": succ}


def interpolate_poses(key_t, key_poses, t, bd_scale=1.0, recenter_c2w=None):
    """data/loader_events.py:133-148: key_t [M], key_poses [M, 3, 4] float64, t [n] -> [n, 4, 4] float32 (fourth row 0 0 0 1)"""
    kt = np.ascontiguousarray(key_t, dtype=np.float64)
    kp = np.ascontiguousarray(np.asarray(key_poses, dtype=np.float64)[:, :3, :4])
    tt = np.ascontiguousarray(t, dtype=np.float64).reshape(-1)
    dp = C.POINTER(C.c_double)
    c44 = None
    if recenter_c2w is not None:
        c = np.asarray(recenter_c2w, dtype=np.float64)
        c44 = np.ascontiguousarray(np.concatenate([c[:3, :4], np.array([[0, 0, 0, 1.0]])], 0))
    out = np.empty((tt.shape[0], 3, 4), np.float32)
    fn = lib().evo_interpolate_poses
    fn.restype = C.c_int
    rc = fn(kt.ctypes.data_as(dp), kp.ctypes.data_as(dp), int(kt.shape[0]), C.c_float(bd_scale), c44.ctypes.data_as(dp) if c44 is not None else None,
            tt.ctypes.data_as(dp), C.c_long(tt.shape[0]), _p(out))
    if rc != 0:
        raise ValueError("interpolate_poses: >= 4 key poses with strictly ascending timestamps")
    full = np.zeros((tt.shape[0], 4, 4), np.float32)
    full[:, :3] = out
    full[:, 3, 3] = 1.0
    return full


def sample_events_track(events, id_to_coords, key_t, key_poses, events_ids, K, bd_scale=1.0, recenter_c2w=None, **kw):
    """sample_events with the per-event poses from interpolate_poses at the events' timestamps (column ncol-3), loader_events.py:280-283"""
    ev = np.asarray(events, dtype=np.float64)
    poses = interpolate_poses(key_t, key_poses, ev[:, -3], bd_scale, recenter_c2w)[:, :3, :4]
    return sample_events(ev, id_to_coords, poses, events_ids, K, **kw)


def image_batch(ray_ids, images, poses, K, pts0_images=None):
    """data/loader.py:325-356 -> the reference's dict (+ 'n_invalid')"""
    ids = np.ascontiguousarray(ray_ids, dtype=np.int64).reshape(-1)
    im, po, Kf = _f(images), _f(np.asarray(poses)[:, :3, :4]), _f(np.asarray(K).reshape(-1))
    p0 = _f(pts0_images) if pts0_images is not None else None
    n = ids.shape[0]
    n_img, H, W = im.shape[:3]
    rays, rx, ry = np.empty((n, 3, 2), np.float32), np.empty((n, 1), np.float32), np.empty((n, 1), np.float32)
    idx, rgb, pout = np.empty((n, 1), np.int64), np.empty((n, 3), np.float32), np.empty((n, 3, 4), np.float32)
    rgb0 = np.empty((n, 3), np.float32) if p0 is not None else None
    fn = lib().evo_image_batch
    fn.restype = C.c_long
    bad = fn(ids.ctypes.data_as(C.POINTER(C.c_longlong)), C.c_long(n), _p(im), _p(p0) if p0 is not None else None, _p(po), int(n_img), int(H), int(W),
             _p(Kf), _p(rays), _p(rx), _p(ry), idx.ctypes.data_as(C.POINTER(C.c_longlong)), _p(rgb), _p(pout), _p(rgb0) if rgb0 is not None else None)
    out = {"rays": rays, "rays_x": rx, "rays_y": ry, "images_idx": idx, "rgbsf": rgb, "poses": pout, "n_invalid": int(bad)}
    if rgb0 is not None:
        out["rgbsf_pts0"] = rgb0
    return out


def event_tables(x, y, t, p, h, w, tmin, tmax, ev_map=None, color_events=True, min_step=0):
    """The array-level part of LLFFEventsDataset.load_event_data (data/loader_events.py:186-255 with utils/events.py:39-66, 72-120): ->
    dict(events [N', 4], id_to_coords, noev_coord_ids, id_to_color_map, events_num_successors, events_with_successor_idx, intcoords)"""
    x, y = _f(x), _f(y)
    t, p = np.ascontiguousarray(t, dtype=np.float64), np.ascontiguousarray(p, dtype=np.float64)
    N, HW = x.shape[0], int(h) * int(w)
    ll = C.POINTER(C.c_longlong)
    dp = C.POINTER(C.c_double)
    ev_ids, noev, i2c = np.empty((max(N, 1),), np.int64), np.empty((HW,), np.int64), np.empty((N + HW, 2), np.float64)
    nn = C.c_long(0)
    fn = lib().evo_event_coord_ids
    fn.restype = C.c_long
    nc = int(fn(_p(x), _p(y), C.c_long(N), int(h), int(w), ev_ids.ctypes.data_as(ll), noev.ctypes.data_as(ll), i2c.ctypes.data_as(dp), C.byref(nn)))
    i2c, noev = i2c[:nc].copy(), noev[:nn.value].copy()
    ev3 = np.empty((max(N, 1), 3), np.float64)
    fn = lib().evo_event_filter
    fn.restype = C.c_long
    n = int(fn(ev_ids.ctypes.data_as(ll), t.ctypes.data_as(dp), p.ctypes.data_as(dp), C.c_long(N), C.c_double(tmin), C.c_double(tmax), ev3.ctypes.data_as(dp)))
    if n < 0:
        raise ValueError("polarities are not {-1, 1} after the normalisation (loader_events.py:206)")
    ev3 = ev3[:n]
    cmap = None
    if color_events:
        cmap = np.zeros((max(nc, 1), 3), np.uint8)
        fn = lib().evo_event_color_map
        fn.restype = C.c_long
        mx, my = (None, None) if ev_map is None else (_f(ev_map[0]), _f(ev_map[1]))
        bad = int(fn(i2c.ctypes.data_as(dp), C.c_long(nc), int(h), int(w), _p(mx) if mx is not None else None, _p(my) if my is not None else None,
                     noev.ctypes.data_as(ll), C.c_long(noev.shape[0]), cmap.ctypes.data_as(C.POINTER(C.c_ubyte))))
        if bad:
            raise ValueError(f"{bad} event coordinates without exactly one colour (loader_events.py:231-234)")
        cmap = cmap[:nc]
    succ, nsucc, _, _ = compute_successor(ev3[:, 0].astype(np.int64), max(nc, 1))
    events = np.concatenate([ev3, np.asarray(succ, dtype=np.float64).reshape(-1, 1)], -1)
    return {"events": events, "id_to_coords": i2c, "noev_coord_ids": noev, "id_to_color_map": cmap, "events_num_successors": np.asarray(nsucc),
            "events_with_successor_idx": np.where(np.asarray(nsucc) > min_step)[0], "intcoords": bool(np.all(i2c.astype(np.int32) == i2c))}


def rbk_warp(rays, r, v, num_motion, use_origin=True, want_transform=False):
    """blurmodel.py:51-82: rays [R,3,2], r/v [R, 3*M] -> new_rays [R, M(+1), 3, 2] (, transforms [R, M(+1), 4, 4])"""
    rays, r, v = _f(rays), _f(r), _f(v)
    R = rays.shape[0]
    P = num_motion + (1 if use_origin else 0)
    out = np.empty((R, P, 3, 2), np.float32)
    tf = np.empty((R, P, 4, 4), np.float32) if want_transform else None
    lib().evo_rbk_warp(_p(rays), _p(r), _p(v), C.c_long(R), num_motion, int(use_origin), _p(out), _p(tf) if tf is not None else None)
    return (out, tf) if want_transform else out


def awp_feature_integration(feat, z, rays_d):
    """feat [N,S,C] (or [R,P,S,C]), z [N,S], rays_d [N,3] -> [N,C] (awp.py:49-77)"""
    feat, z, rays_d = _f(feat), _f(z), _f(rays_d)
    S, Cc = feat.shape[-2], feat.shape[-1]
    N = z.shape[0]
    out = np.empty((N, Cc), np.float32)
    lib().evo_awp_feature_integration(_p(feat), _p(z), _p(rays_d), C.c_long(N), S, Cc, _p(out))
    return out


def awp_sample_embed(x, weights, biases):
    """x [n, in], weights[l] [W, in_l], biases[l] [W] -> [n, W] (awp.py:98-100: depth x Linear + ReLU)"""
    x = _f(x).reshape(-1, np.asarray(weights[0]).shape[1])
    ws, bs = [_f(w) for w in weights], [_f(b) for b in biases]
    fp = C.POINTER(C.c_float)
    wa = (fp * len(ws))(*[w.ctypes.data_as(fp) for w in ws])
    ba = (fp * len(bs))(*[b.ctypes.data_as(fp) for b in bs])
    n, width = x.shape[0], ws[0].shape[0]
    if x.shape[1] > 512 or width > 512:
        raise ValueError("evo_awp_sample_embed: widths up to 512")
    out = np.empty((n, width), np.float32)
    lib().evo_awp_sample_embed(_p(x), wa, ba, C.c_long(n), x.shape[1], width, len(ws), _p(out))
    return out


def mam_local(x_local, W, b, v, P):
    """x_local [R P, S, C], MAM.linear W [M, C] / b [M], Corr.line_conv_att v [M] -> curver_inter [R, M, P], curves_intra [R, M, S]
    (mam.py:72-74, 29-33 as written)"""
    x, W, b, v = _f(x_local), _f(W), _f(b), _f(np.asarray(v).reshape(-1))
    RP, S, Cc = x.shape
    R, M = RP // P, W.shape[0]
    inter, intra = np.empty((R, M, P), np.float32), np.empty((R, M, S), np.float32)
    lib().evo_mam_local(_p(x), _p(W), _p(b), _p(v), C.c_long(R), P, S, Cc, M, _p(inter), _p(intra))
    return inter, intra


def awp_per_ray(h, view, inter, intra, sd, training=True, eps=1e-5):
    """The adaptive weight proposal behind its per-sample part (awp.py:104-117, mam.py:35-53 as written).  h [R, P, Ws] integrated features,
    view [R, VC] = view_embedded (view_feature + direction encoding, awp.py:89-95), inter [R, Cm, P] / intra [R, Cm, S] = mam_local's outputs,
    sd: the module's state dict (reference names).  -> (out [R, P], batch mean [Cm], unbiased batch variance [Cm])"""
    h, view, inter, intra = _f(h), _f(view), _f(inter), _f(intra)
    R, P, Ws = h.shape
    Cm, S, VC = inter.shape[1], intra.shape[2], view.shape[1]
    g = lambda k: _f(np.asarray(sd[k]).reshape(np.asarray(sd[k]).shape[0], -1) if np.asarray(sd[k]).ndim > 1 else np.asarray(sd[k]))
    n_mot = len([k for k in sd if k.startswith("motion_feature_embed_layer.") and k.endswith(".weight")])
    mw, mb = [g(f"motion_feature_embed_layer.{l}.weight") for l in range(n_mot)], [g(f"motion_feature_embed_layer.{l}.bias") for l in range(n_mot)]
    if P > 64 or Cm > 512:
        raise ValueError("evo_awp_per_ray: P <= 64, Cm <= 512")
    fp = C.POINTER(C.c_float)
    wa, ba = (fp * n_mot)(*[w.ctypes.data_as(fp) for w in mw]), (fp * n_mot)(*[b.ctypes.data_as(fp) for b in mb])
    cv = {k: g(f"MAM.Corr.{k}.weight") for k in ("conva", "convb", "convc", "convn", "convl")}
    cd, bw, bb = g("MAM.Corr.convd.0.weight"), g("MAM.Corr.convd.1.weight"), g("MAM.Corr.convd.1.bias")
    rm, rv = g("MAM.Corr.convd.1.running_mean"), g("MAM.Corr.convd.1.running_var")
    ww, wb = g("w_linear.weight"), g("w_linear.bias")
    out, stats = np.empty((R, P), np.float32), np.empty((2 * Cm,), np.float32)
    lib().evo_awp_per_ray(_p(h), _p(view), _p(inter), _p(intra), wa, ba, n_mot, _p(cv["conva"]), _p(cv["convb"]), _p(cv["convc"]), _p(cv["convn"]),
                          _p(cv["convl"]), _p(cd), _p(bw), _p(bb), _p(rm), _p(rv), C.c_float(eps), int(bool(training)), _p(ww), _p(wb), C.c_long(R), P, S,
                          Ws, VC, Cm, _p(out), _p(stats))
    return out, stats[:Cm], stats[Cm:]


def crf_forward(crf: Crf, x, feat=None, skip_learn=False):
    x = _f(x)
    n = x.shape[0]
    per_ch = 0
    if feat is not None:
        feat = _f(feat)
        per_ch = int(feat.ndim == 3)
    out = np.empty_like(x)
    lib().evo_crf_forward(C.byref(crf.s), _p(x), _p(feat), per_ch, int(skip_learn), C.c_long(n), _p(out))
    return out


def luma(x, standard="rec601"):
    x = _f(x)
    out = np.empty((x.shape[0], 1), np.float32)
    lib().evo_luma(_p(x), C.c_long(x.shape[0]), {"rec601": 0, "rec709": 1, "avg": 2}[standard], _p(out))
    return out


def mse(a, b):
    a, b = _f(a), _f(b)
    return lib().evo_mse(_p(a), _p(b), C.c_long(a.size))


def egm_loss(ls, le, bii, color_mask=None, color_weight=None):
    ls, le, bii = _f(ls), _f(le), _f(bii)
    n, Cc = ls.shape
    cm = np.ascontiguousarray(color_mask, dtype=np.uint8) if color_mask is not None else None
    cw = _f(color_weight) if color_weight is not None else None
    return lib().evo_egm_loss(_p(ls), _p(le), _p(bii), C.c_long(n), Cc,
                              cm.ctypes.data_as(C.POINTER(C.c_ubyte)) if cm is not None else None, _p(cw))


def bii_image(x, y, p, w, h, c_pos, c_neg, interpolate=True):
    x, y = _f(x), _f(y)
    p = np.ascontiguousarray(p, dtype=np.int8)
    img = np.empty((h, w), np.float32)
    lib().evo_bii_image(_p(x), _p(y), p.ctypes.data_as(C.POINTER(C.c_byte)), C.c_long(x.shape[0]), w, h,
                        C.c_float(c_pos), C.c_float(c_neg), int(interpolate), _p(img))
    return img


def inner_double_integral(bii):
    bii = _f(bii)
    steps = bii.shape[0] + 1
    npix = int(np.prod(bii.shape[1:]))
    out = np.empty((steps,) + bii.shape[1:], np.float32)
    lib().evo_inner_double_integral(_p(bii), steps, C.c_long(npix), _p(out))
    return out


def deblur_double_integral(blurry, bii):
    blurry, bii = _f(blurry), _f(bii)
    out = np.empty_like(blurry)
    lib().evo_deblur_double_integral(_p(blurry), _p(bii), bii.shape[0] + 1, C.c_long(blurry.size), _p(out))
    return out


def tv_loss(x):
    x = _f(x)
    _, Cc, H, W = x.shape
    return lib().evo_tv_loss(_p(x), Cc, H, W)


def tv_loss_app(sd, prefix=""):
    """VoxelNeRFBase.TV_loss_app (voxnerf.py:126-130): sum over the three axes of reg(plane) 1e-2 + reg(line) 1e-3"""
    tot = 0.0
    for i in range(3):
        tot += tv_loss(np.asarray(sd[f"{prefix}app_plane.{i}"])) * 1e-2 + tv_loss(np.asarray(sd[f"{prefix}app_line.{i}"])) * 1e-3
    return tot


def train_forward(coarse: Voxel, fine: Voxel, cfg, new_rays, weight, img_embed, awp_sd, sd_levels, ccw_fine_scale=0.05, ray_dir_freq=2):
    """The training branch of NeRFAll.forward behind the blur kernel (networks/renderer.py:303-376, kernel_type RBK, use_awp, mode
    c2f, N_importance > 0) composed from the oracle's functions on the kernel's recorded outputs:
      render (:306) of new_rays [R,P,3,2] flattened -> rgb [R P,3], rgb0, depth_feature [R P,S,G], z_vals, NDC rays_d (:464-465);
      awpnet (:314-315, awp.py:79-117) = sample embedding -> feature integration -> MAM per-sample sums -> per-ray remainder, with
        view_embedded = [img_embed | PE(first sub-exposure's normalised NDC direction)] (awp.py:87-95);
      ccw_fine + ccw_fine * 0.05, renormalised (:316-317); rbk_weighted_sum with weight1 and with ccw_fine (:327-330, blurmodel.py:112-127);
      rgb1 = the weighted extras['rgb0'] (:341); TV = (coarse + fine TV_loss_app) * 5 (:361-365); the pts0 tensors (:371-376).
    awp_sd: the AdaptiveWeightProposal's state dict (numpy, reference names); sd_levels: the model's state dict (for the TV term).
    -> dict(rgb, rgb1, rgb_awp, stage1_rgb_pts0, stage1_rgb1_pts0, tv, ccw_fine, bn_mean, bn_var, render=<the render's dict>)"""
    new_rays, weight = _f(new_rays), _f(weight)
    R, P = weight.shape
    flat = new_rays.reshape(R * P, 3, 2)
    res = render_c2f(coarse, fine, cfg, flat, want_feature=True)
    rb = ray_batch(cfg, flat)
    rays_d = rb[:, 3:6].copy()
    S = res["z_vals"].shape[1]
    n_emb = len([k for k in awp_sd if k.startswith("sample_feature_embed_layer.") and k.endswith(".weight")])
    h_local = awp_sample_embed(res["feature"].reshape(R * P * S, -1), [awp_sd[f"sample_feature_embed_layer.{l}.weight"] for l in range(n_emb)],
                               [awp_sd[f"sample_feature_embed_layer.{l}.bias"] for l in range(n_emb)]).reshape(R * P, S, -1)
    h = awp_feature_integration(h_local, res["z_vals"], rays_d).reshape(R, P, -1)
    d0 = rays_d.reshape(R, P, 3)[:, 0]
    enc = embed(d0 / np.linalg.norm(d0, axis=-1, keepdims=True), ray_dir_freq)
    view = enc if img_embed is None else np.concatenate([_f(img_embed), enc], -1)
    inter, intra = mam_local(h_local, awp_sd["MAM.linear.weight"], awp_sd["MAM.linear.bias"], awp_sd["MAM.Corr.line_conv_att.weight"], P)
    ccw, mean, var = awp_per_ray(h, view, inter, intra, awp_sd, training=True)
    ccw = ccw + ccw * np.float32(ccw_fine_scale)
    ccw = ccw / ccw.sum(-1, keepdims=True)
    out = dict(rgb=weighted_sum(res["rgb"], weight), rgb1=weighted_sum(res["rgb0"], weight), rgb_awp=weighted_sum(res["rgb"], ccw),
               stage1_rgb_pts0=res["rgb"].reshape(R, P, 3)[:, 0].copy(), stage1_rgb1_pts0=res["rgb0"].reshape(R, P, 3)[:, 0].copy(),
               ccw_fine=ccw, bn_mean=mean, bn_var=var, render=res)
    out["tv"] = (tv_loss_app(sd_levels, "mlp_coarse.") + tv_loss_app(sd_levels, "mlp_fine.")) * 5
    return out


def num_threads():
    return lib().evo_num_threads()
