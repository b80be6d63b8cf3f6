"""Mirror of the reference ``networks/nerf.py`` NeRF module on libevdnerf.so.

The MLP (mlpforward + eval, nerf.py:46-72,131-162) runs as ONE fused HIP kernel that also computes
pts = o + d z and both positional encodings, so this class takes the packed ray batch + z_vals instead
of ``pts, viewdirs, embed_fn, embeddirs_fn``; ``raw2outputs`` keeps the reference signature and 6-tuple.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib as L


def _np32(v):
    if isinstance(v, torch.Tensor):
        v = v.detach().cpu().numpy()
    return np.ascontiguousarray(v, dtype=np.float32)


class _Raw2Outputs(torch.autograd.Function):
    """raw2outputs as an autograd node (C = 4, three colours): forward = evd_raw2outputs, backward = evd_raw2outputs_bwd
    (the forward quantities are recomputed in the backward kernel; z / rays_d get no gradient, as in the reference where
    the sample positions are detached)."""

    @staticmethod
    def forward(ctx, raw, z, rd, noise, sigma_ch, rgb_ch0, rgb_act, sigma_act, white_bkgd, thr):
        R, S, Cc = raw.shape
        dev = raw.device
        rgb = torch.empty((R, 3), dtype=torch.float32, device=dev)
        dens = torch.empty((R, S - 1), dtype=torch.float32, device=dev)
        acc = torch.empty((R,), dtype=torch.float32, device=dev)
        wts = torch.empty((R, S), dtype=torch.float32, device=dev)
        depth = torch.empty((R,), dtype=torch.float32, device=dev)
        L.check(L.lib().evd_raw2outputs(L.ptr(raw), L.ptr(z), L.ptr(rd), rd.shape[-1], R, S, Cc, sigma_ch, rgb_ch0, 3, rgb_act, sigma_act,
                                        int(bool(white_bkgd)), thr, L.ptr(noise), L.ptr(rgb), L.ptr(dens), L.ptr(acc), L.ptr(wts),
                                        L.ptr(depth), None, 0, None, L.stream_ptr()), "evd_raw2outputs")
        ctx.save_for_backward(raw, z, rd, noise if noise is not None else torch.empty(0, device=dev))
        ctx.cfg = (sigma_ch, rgb_ch0, rgb_act, sigma_act, int(bool(white_bkgd)), thr, noise is not None)
        ctx.mark_non_differentiable(dens)
        return rgb, dens, acc, wts, depth

    @staticmethod
    def backward(ctx, g_rgb, g_dens, g_acc, g_wts, g_depth):
        raw, z, rd, noise = ctx.saved_tensors
        sigma_ch, rgb_ch0, rgb_act, sigma_act, white, thr, has_noise = ctx.cfg
        R, S, Cc = raw.shape
        c = lambda g: g.contiguous().float() if g is not None else None
        d_raw = torch.empty_like(raw)
        L.check(L.lib().evd_raw2outputs_bwd(L.ptr(raw), L.ptr(z), L.ptr(rd), rd.shape[-1], R, S, Cc, sigma_ch, rgb_ch0, 3, rgb_act, sigma_act,
                                            white, thr, L.ptr(noise) if has_noise else None, L.ptr(c(g_rgb)), L.ptr(c(g_depth)),
                                            L.ptr(c(g_acc)), L.ptr(c(g_wts)), L.ptr(d_raw), L.stream_ptr()), "evd_raw2outputs_bwd")
        return (d_raw,) + (None,) * 9


class NeRF:
    """One reference ``NeRF`` (D x W MLP, skip, view branch) with packed MFMA weight streams on the GPU."""

    def __init__(self, state_dict, prefix="", D=8, W=256, multires=10, multires_views=4, skips=(4,),
                 rgb_activate="sigmoid", sigma_activate="relu", render_rmnearplane=0,
                 extract_feature="after_linear", composite_feature=True, precision="f16x3"):
        self.D, self.W = D, W
        self.multires, self.multires_views = multires, multires_views
        self.rgb_activate, self.sigma_activate = rgb_activate, sigma_activate
        self.render_rmnearplane = render_rmnearplane
        self.extract_feature, self.composite_feature = extract_feature, composite_feature
        self.precision = precision
        self.training = False
        g = lambda k: _np32(state_dict[prefix + k]) if (prefix + k) in state_dict else None
        keep = []
        d = L.NerfDesc()
        d.D, d.W, d.multires, d.multires_views = D, W, multires, multires_views
        d.skip = skips[0] if len(skips) else -1
        d.rgb_act, d.sigma_act, d.rmnear = L.ACT[rgb_activate], L.ACT[sigma_activate], float(render_rmnearplane)
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None
        for i in range(D):
            w, b = g(f"pts_linears.{i}.weight"), g(f"pts_linears.{i}.bias")
            if w is None or b is None:
                raise L.EvdError(f"state dict lacks {prefix}pts_linears.{i}")
            keep += [w, b]
            d.pts_w[i], d.pts_b[i] = fp(w), fp(b)
        for name, key in (("views", "views_linears.0"), ("feature", "feature_linear"), ("alpha", "alpha_linear"),
                          ("rgb", "rgb_linear")):
            w, b = g(key + ".weight"), g(key + ".bias")
            keep += [w, b]
            setattr(d, name + "_w", fp(w))
            setattr(d, name + "_b", fp(b))
        h = C.c_void_p()
        L.check(L.lib().evd_nerf_create(C.byref(d), C.byref(h)), "evd_nerf_create")
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and L is not None and getattr(L, "lib", None) is not None:     # module globals may be gone at interpreter shutdown
            L.lib().evd_nerf_destroy(h)
            self._h = None

    @property
    def handle(self):
        return self._h

    def train(self, mode=True):
        self.training = mode
        return self

    def stream_bytes(self, precision=None):
        return int(L.lib().evd_nerf_stream_bytes(self._h, L.PREC[precision or self.precision]))

    # nerf.py:46-72 + :131-162 (fused with renderer.py:180 and embedding.py:88-98)
    def mlpforward(self, ray_batch, z_vals, want_feature=False, precision=None):
        rb = ray_batch.contiguous().float()
        z = z_vals.contiguous().float()
        if rb.shape[1] != 11:
            raise L.EvdError("mlpforward needs the 11-column ray batch (use_viewdirs=True)")
        R, S = z.shape
        raw = torch.empty((R, S, 4), dtype=torch.float32, device=z.device)
        feat = torch.empty((R, S, self.W), dtype=torch.float32, device=z.device) if want_feature else None
        kind = 2 if self.extract_feature == "before_linear" else 1
        L.check(L.lib().evd_nerf_mlp(self._h, L.PREC[precision or self.precision], L.ptr(rb), L.ptr(z), R, S,
                                     L.ptr(raw), L.ptr(feat), kind, L.stream_ptr()), "evd_nerf_mlp")
        return raw, feat

    # Training forward: the same raw plus the activation store evd_nerf_mlp_backward consumes (f16 / bf16, 8 x 256 network)
    def mlpforward_train(self, ray_batch, z_vals, precision=None):
        rb = ray_batch.contiguous().float()
        z = z_vals.contiguous().float()
        R, S = z.shape
        raw = torch.empty((R, S, 4), dtype=torch.float32, device=z.device)
        nb = int(L.lib().evd_nerf_train_store_bytes(R * S))
        store = torch.empty((nb,), dtype=torch.uint8, device=z.device)
        L.check(L.lib().evd_nerf_mlp_train(self._h, L.PREC[precision or self.precision], L.ptr(rb), L.ptr(z), R, S, L.ptr(raw),
                                            L.ptr(store), nb, L.stream_ptr()), "evd_nerf_mlp_train")
        return raw, store

    # Backward of mlpforward_train: d_raw [R,S,4] -> {state-dict key: gradient} (what autograd gives for nerf.py:46-72)
    def mlp_backward(self, d_raw, store, precision=None):
        g = d_raw.contiguous().float()
        R, S = g.shape[:2]
        dev = g.device
        shapes = {f"pts_linears.{i}": (self.W, 63 if i == 0 else (self.W + 63 if i == 5 else self.W)) for i in range(self.D)}
        shapes.update({"views_linears.0": (self.W // 2, self.W + 27), "feature_linear": (self.W, self.W),
                       "alpha_linear": (1, self.W), "rgb_linear": (3, self.W // 2)})
        out = {}
        gs = L.NerfGrads()
        for key, (o, i) in shapes.items():
            w = torch.zeros((o, i), dtype=torch.float32, device=dev)
            b = torch.zeros((o,), dtype=torch.float32, device=dev)
            out[key + ".weight"], out[key + ".bias"] = w, b
            if key.startswith("pts_linears."):
                k = int(key.split(".")[1])
                gs.pts_w[k], gs.pts_b[k] = L.ptr(w), L.ptr(b)
            else:
                name = {"views_linears.0": "views", "feature_linear": "feature", "alpha_linear": "alpha", "rgb_linear": "rgb"}[key]
                setattr(gs, name + "_w", L.ptr(w))
                setattr(gs, name + "_b", L.ptr(b))
        nb = int(L.lib().evd_nerf_backward_workspace_bytes())
        ws = torch.empty((nb,), dtype=torch.uint8, device=dev)
        L.check(L.lib().evd_nerf_mlp_backward(self._h, L.PREC[precision or self.precision], L.ptr(g), R, S, L.ptr(store), store.numel(),
                                               C.byref(gs), L.ptr(ws), nb, L.stream_ptr()), "evd_nerf_mlp_backward")
        return out

    # nerf.py:74-129; returns the reference 6-tuple (rgb_map, density, acc_map, weights, depth_map, feature_map)
    def raw2outputs(self, raw, z_vals, rays_d, feature=None, raw_noise_std=0, white_bkgd=False, pytest=False,
                    noise=None):
        raw = raw.contiguous().float()
        z = z_vals.contiguous().float()
        rd = rays_d.contiguous().float()
        R, S, Cc = raw.shape
        dev = raw.device
        if raw_noise_std > 0. and noise is None:
            noise = torch.randn((R, S - 1), dtype=torch.float32, device=dev) * raw_noise_std
        rgb = torch.empty((R, 3), dtype=torch.float32, device=dev)
        dens = torch.empty((R, S - 1), dtype=torch.float32, device=dev)
        acc = torch.empty((R,), dtype=torch.float32, device=dev)
        wts = torch.empty((R, S), dtype=torch.float32, device=dev)
        depth = torch.empty((R,), dtype=torch.float32, device=dev)
        F = feature.shape[-1] if feature is not None else 0
        fmap = torch.empty((R, F), dtype=torch.float32, device=dev) if feature is not None else None
        ft = feature.contiguous().float() if feature is not None else None
        thr = float(self.render_rmnearplane) / 128.0 if (not self.training and self.render_rmnearplane > 0) else 0.0
        nz = noise.contiguous().float() if noise is not None else None
        if raw.requires_grad and torch.is_grad_enabled() and feature is None and Cc == 4:
            # training through the compositing scan: autograd node backed by evd_raw2outputs_bwd
            rgb, dens, acc, wts, depth = _Raw2Outputs.apply(raw, z, rd, nz, 3, 0, L.ACT[self.rgb_activate], L.ACT[self.sigma_activate],
                                                            white_bkgd, thr)
            return rgb, dens, acc, wts, depth, None
        L.check(L.lib().evd_raw2outputs(L.ptr(raw), L.ptr(z), L.ptr(rd), rd.shape[-1], R, S, Cc, 3, 0, 3,
                                        L.ACT[self.rgb_activate], L.ACT[self.sigma_activate], int(bool(white_bkgd)),
                                        thr, L.ptr(nz), L.ptr(rgb), L.ptr(dens), L.ptr(acc), L.ptr(wts), L.ptr(depth),
                                        L.ptr(ft), F, L.ptr(fmap), L.stream_ptr()), "evd_raw2outputs")
        return rgb, dens, acc, wts, depth, fmap

    # nerf.py:164-175; 5-tuple (rgb_map, depth_map, acc_map, weights, feature_map)
    def forward(self, ray_batch, z_vals, raw_noise_std=0., white_bkgd=False, is_train=False, want_feature=False):
        raw, feature = self.mlpforward(ray_batch, z_vals, want_feature=want_feature)
        rays_d = ray_batch[:, 3:6]
        if self.composite_feature:
            rgb, _, acc, wts, depth, fmap = self.raw2outputs(raw, z_vals, rays_d, feature, raw_noise_std, white_bkgd)
        else:
            rgb, _, acc, wts, depth, _ = self.raw2outputs(raw, z_vals, rays_d, None, raw_noise_std, white_bkgd)
            fmap = feature
        return rgb, depth, acc, wts, fmap

    __call__ = forward
