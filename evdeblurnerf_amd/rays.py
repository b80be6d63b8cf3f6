"""Mirror of the reference ``utils/rays.py`` (get_rays :8, get_rays_pix :25, get_ndc_rays :104,
sample_pdf :149) on top of libevdnerf.so. Same names, argument meaning and return conventions;
tensors are float32 CUDA tensors."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib as L


def _k9(K):
    k = K.detach().cpu().numpy() if isinstance(K, torch.Tensor) else np.asarray(K)
    return np.ascontiguousarray(k, dtype=np.float32).reshape(9)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def get_rays(H, W, K, c2w, add_halfpix=True):
    """utils/rays.py:8-22 -> (rays_o [H,W,3], rays_d [H,W,3])."""
    k = _k9(K)
    pose = np.ascontiguousarray(c2w.detach().cpu().numpy() if isinstance(c2w, torch.Tensor) else c2w,
                                dtype=np.float32)[:3, :4].copy()
    dev = c2w.device if isinstance(c2w, torch.Tensor) and c2w.is_cuda else torch.device("cuda")
    o = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
    d = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
    L.check(L.lib().evd_get_rays(H, W, _fp(k), _fp(pose), int(bool(add_halfpix)), L.ptr(o), L.ptr(d), L.stream_ptr()), "evd_get_rays")
    return o, d


def get_rays_pix(coords, K, c2ws, add_halfpix=True):
    """utils/rays.py:25-36: coords [n,2] (x,y), c2ws [n,3,4] -> (rays_o [n,3], rays_d [n,3]).  add_halfpix=False: the coordinates are
    sub-pixel positions already (the event loader passes integer_coords, data/loader_events.py:290-293)."""
    k = _k9(K)
    coords = coords.contiguous().float()
    c2ws = c2ws[..., :3, :4].contiguous().float()
    n = coords.shape[0]
    o = torch.empty((n, 3), dtype=torch.float32, device=coords.device)
    d = torch.empty((n, 3), dtype=torch.float32, device=coords.device)
    L.check(L.lib().evd_get_rays_pix(L.ptr(coords), _fp(k), L.ptr(c2ws), n, int(bool(add_halfpix)), L.ptr(o), L.ptr(d), L.stream_ptr()),
            "evd_get_rays_pix")
    return o, d


def get_ndc_rays(H, W, focal, near, rays_o, rays_d):
    """utils/rays.py:104-145."""
    sh = rays_o.shape
    o = rays_o.reshape(-1, 3).contiguous().float()
    d = rays_d.reshape(-1, 3).contiguous().float()
    oo = torch.empty_like(o)
    od = torch.empty_like(d)
    L.check(L.lib().evd_ndc_rays(int(H), int(W), float(focal), float(near), L.ptr(o), L.ptr(d), o.shape[0],
                                 L.ptr(oo), L.ptr(od), L.stream_ptr()), "evd_ndc_rays")
    return oo.reshape(sh), od.reshape(sh)


def sample_pdf_merge(z_vals, weights, N_samples, det=False, u=None, want_order=False):
    """Fused renderer.py:200-205/230-234 + :250: returns (z_samples, z_merged, order|None, z_std)."""
    z = z_vals.contiguous().float()
    w = weights.contiguous().float()
    R, S = z.shape
    if not det and u is None:
        u = torch.rand((R, N_samples), dtype=torch.float32, device=z.device)
    zs = torch.empty((R, N_samples), dtype=torch.float32, device=z.device)
    zm = torch.empty((R, S + N_samples), dtype=torch.float32, device=z.device)
    order = torch.empty((R, S + N_samples), dtype=torch.int32, device=z.device) if want_order else None
    zstd = torch.empty((R,), dtype=torch.float32, device=z.device)
    uu = u.contiguous().float() if u is not None else None
    L.check(L.lib().evd_sample_pdf_merge(L.ptr(z), L.ptr(w), R, S, N_samples, int(bool(det)), L.ptr(uu),
                                         L.ptr(zs), L.ptr(zm), L.ptr(order), L.ptr(zstd), L.stream_ptr()),
            "evd_sample_pdf_merge")
    return zs, zm, order, zstd


def sample_pdf(bins, weights, N_samples, det=False, pytest=False, u=None):
    """utils/rays.py:149-193 for the only way the reference calls it (renderer.py:200-201,230-231):
    ``bins`` are the mid-points of a z_vals row and ``weights`` its ``[...,1:-1]`` slice. The C ABI takes
    z_vals / full weights, so this wrapper rebuilds them: z_vals is recovered from the mid-points up to its
    two end samples, which sample_pdf never reads (only bins enter), so any extension gives the same result.
    """
    b = bins.contiguous().float()
    R, nb = b.shape
    S = nb + 1
    # construct a z row whose midpoints are `bins`: z[0] free, z[i+1] = 2 b[i] - z[i]
    z = torch.empty((R, S), dtype=torch.float32, device=b.device)
    z[:, 0] = b[:, 0]
    for i in range(nb):
        z[:, i + 1] = 2 * b[:, i] - z[:, i]
    mid = 0.5 * (z[:, 1:] + z[:, :-1])
    if not torch.equal(mid, b):
        raise L.EvdError("sample_pdf: bins are not exactly representable as mid-points; call sample_pdf_merge "
                         "with z_vals instead (the path the renderer uses)")
    w = torch.zeros((R, S), dtype=torch.float32, device=b.device)
    w[:, 1:-1] = weights
    return sample_pdf_merge(z, w, N_samples, det=det, u=u)[0]


def rbk_warp(rays, r, v, num_motion, use_origin=True, return_transform=False):
    """RigidBlurringModel.rbk_warp (networks/dpnerf/blurmodel.py:51-82): rays [R,3,2], r / v [R, 3*num_motion] (the r_linear /
    v_linear outputs) -> new_rays [R, num_motion (+1), 3, 2] (and the [.., 4, 4] rigid transforms)."""
    rr = rays.contiguous().float()
    R = rr.shape[0]
    P = num_motion + (1 if use_origin else 0)
    out = torch.empty((R, P, 3, 2), dtype=torch.float32, device=rr.device)
    tf = torch.empty((R, P, 4, 4), dtype=torch.float32, device=rr.device) if return_transform else None
    L.check(L.lib().evd_rbk_warp(L.ptr(rr), L.ptr(r.contiguous().float()), L.ptr(v.contiguous().float()), R, int(num_motion), int(bool(use_origin)),
                                 L.ptr(out), L.ptr(tf), L.stream_ptr()), "evd_rbk_warp")
    return (out, tf) if return_transform else out
