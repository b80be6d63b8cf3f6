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
    (the forward quantities are recomputed in the backward kernel; z gets no gradient, as in the reference where the sample
    positions are detached; rays_d gets the one through dists = dz |rays_d| when it requires grad, relu density only)."""

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
        ctx.set_materialize_grads(False)       # unused outputs (depth, acc, often the weights) arrive as None, not as freshly zeroed tensors
        return rgb, dens, acc, wts, depth

    @staticmethod
    def backward(ctx, g_rgb, g_dens, g_acc, g_wts, g_depth):
        raw, z, rd, noise = ctx.saved_tensors
        sigma_ch, rgb_ch0, rgb_act, sigma_act, white, thr, has_noise = ctx.cfg
        R, S, Cc = raw.shape
        c = lambda g: g.contiguous().float() if g is not None else None
        d_raw = torch.empty_like(raw)
        # the ray directions enter through dists * |d| only; their gradient (the blur batch's warped rays) is a per-ray sum the kernel
        # has in registers (eight small tensor operations per level before)
        d_rd = torch.empty_like(rd) if ctx.needs_input_grad[2] else None
        L.check(L.lib().evd_raw2outputs_bwd_rays(L.ptr(raw), L.ptr(z), L.ptr(rd), rd.shape[-1], R, S, Cc, sigma_ch, rgb_ch0, 3, rgb_act, sigma_act,
                                                 white, thr, L.ptr(noise) if has_noise else None, L.ptr(c(g_rgb)), L.ptr(c(g_depth)),
                                                 L.ptr(c(g_acc)), L.ptr(c(g_wts)), L.ptr(d_raw), L.ptr(d_rd), rd.shape[-1], L.stream_ptr()),
                "evd_raw2outputs_bwd_rays")
        return (d_raw, None, d_rd) + (None,) * 7


class _NerfMLP(torch.autograd.Function):
    """raw = NeRF.mlpforward(rays, z; params) with the hand-written backward kernels (evd_nerf_mlp_train / _backward)"""

    @staticmethod
    def forward(ctx, flat, ray_batch, z_vals, net, precision):
        raw, store = net.mlpforward_train(ray_batch, z_vals, precision)
        ctx.net, ctx.precision, ctx.store = net, precision, store
        ctx.rb, ctx.z = ray_batch, z_vals
        ctx.accum = getattr(flat, "_evd_accum", None)          # in-place gradient accumulation (renderer._FlatParams), opt-in
        return raw

    @staticmethod
    def backward(ctx, d_raw):
        L.note_backward()
        d_rb = None
        acc = ctx.accum() if (ctx.accum is not None and ctx.needs_input_grad[0] and not torch.is_grad_enabled()) else None
        if ctx.needs_input_grad[1]:
            # rays: pts = o + d z enters PE(pts) (layer 0 and the skip layer), the view direction PE(dirs); z is a constant
            rb, z = ctx.rb.detach().contiguous().float(), ctx.z.detach().contiguous().float()
            pts = (rb[:, None, 0:3] + rb[:, None, 3:6] * z[..., None]).contiguous()
            g, d_pts, d_dirs = ctx.net.mlp_backward_flat(d_raw, ctx.store, ctx.precision, pts=pts, ray_batch=rb, accumulate_into=acc)
            d_rb = torch.zeros_like(rb)
            d_pts = d_pts.reshape(pts.shape)
            d_rb[:, 0:3] = d_pts.sum(1)
            d_rb[:, 3:6] = (d_pts * z[..., None]).sum(1)
            d_rb[:, 8:11] = d_dirs.reshape(pts.shape).sum(1)
        else:
            g = ctx.net.mlp_backward_flat(d_raw, ctx.store, ctx.precision, accumulate_into=acc)
        ctx.store = None
        return g, d_rb, None, None, None


class NeRF:
    """One reference ``NeRF`` (D x W MLP, skip, view branch) with packed MFMA weight streams on the GPU."""

    def __init__(self, state_dict, prefix="", D=8, W=256, multires=10, multires_views=4, skips=(4,),
                 rgb_activate="sigmoid", sigma_activate="relu", render_rmnearplane=0,
                 extract_feature="after_linear", composite_feature=True, precision="f16x3", use_viewdirs=True):
        """use_viewdirs=False (nerf.py:41-44,158-160): `output_linear` [output_ch, W] replaces the view branch; the ray batch then has
        8 columns (renderer.py:443-446).  Inference only, every mode but f16c.  As in the reference, such a network has no
        "after_linear" feature (nerf.py:159 asserts)."""
        self.D, self.W = D, W
        self.use_viewdirs = bool(use_viewdirs)
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
        if self.use_viewdirs:
            for name, key in (("views", "views_linears.0"), ("feature", "feature_linear"), ("alpha", "alpha_linear"),
                              ("rgb", "rgb_linear")):
                w, b = g(key + ".weight"), g(key + ".bias")
                keep += [w, b]
                setattr(d, name + "_w", fp(w))
                setattr(d, name + "_b", fp(b))
        else:
            w, b = g("output_linear.weight"), g("output_linear.bias")
            if w is None or b is None:
                raise L.EvdError(f"state dict lacks {prefix}output_linear (use_viewdirs=False)")
            keep += [w, b]
            d.output_w, d.output_b, d.output_ch = fp(w), fp(b), int(w.shape[0])
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
        ncol = 11 if self.use_viewdirs else 8
        if rb.shape[1] != ncol:
            raise L.EvdError(f"mlpforward needs the {ncol}-column ray batch (use_viewdirs={self.use_viewdirs})")
        if want_feature and not self.use_viewdirs and self.extract_feature == "after_linear":
            raise L.EvdError('a use_viewdirs=False network has no "after_linear" feature (nerf.py:159)')
        R, S = z.shape
        raw = torch.empty((R, S, 4), dtype=torch.float32, device=z.device)
        feat = torch.empty((R, S, self.W), dtype=torch.float32, device=z.device) if want_feature else None
        kind = 2 if self.extract_feature == "before_linear" else 1
        L.check(L.lib().evd_nerf_mlp(self._h, L.PREC[precision or self.precision], L.ptr(rb), L.ptr(z), R, S,
                                     L.ptr(raw), L.ptr(feat), kind, L.stream_ptr()), "evd_nerf_mlp")
        return raw, feat

    # Training forward: the same raw plus the activation store evd_nerf_mlp_backward consumes (f16 / bf16, or f16x3 = the float32-grade
    # mode with (hi, lo) fragments; 8 x 256 network)
    def mlpforward_train(self, ray_batch, z_vals, precision=None):
        if not self.use_viewdirs:
            raise L.EvdError("the training path is not built for use_viewdirs=False networks")
        rb = ray_batch.contiguous().float()
        z = z_vals.contiguous().float()
        R, S = z.shape
        raw = torch.empty((R, S, 4), dtype=torch.float32, device=z.device)
        nb = int(L.lib().evd_nerf_train_store_bytes_prec(L.PREC[precision or self.precision], R * S))
        if nb == 0 and R * S > 0:
            raise L.EvdError(f"the training path is built for precision f16 / bf16 / f16x3 / f16c / f16m, not {precision or self.precision}")
        store = torch.empty((nb,), dtype=torch.uint8, device=z.device)
        L.check(L.lib().evd_nerf_mlp_train(self._h, L.PREC[precision or self.precision], L.ptr(rb), L.ptr(z), R, S, L.ptr(raw),
                                            L.ptr(store), nb, L.stream_ptr()), "evd_nerf_mlp_train")
        return raw, store

    # ---- training path (f16 / bf16 on the 8 x 256 network) ------------------------------------------------------------
    # Parameters as ONE flat float32 tensor in the library's canonical order (include/evdnerf.h: evd_nerf_param_blocks).
    def param_blocks(self):
        """[(state-dict key, shape, offset)] of the flat parameter tensor"""
        if getattr(self, "_blocks", None) is None:
            nb = 2 * self.D + 8
            off = (C.c_long * (nb + 1))()
            if L.lib().evd_nerf_param_blocks(self._h, off, nb + 1) != nb:
                raise L.EvdError("evd_nerf_param_blocks: unexpected parameter block count")
            W, D = self.W, self.D
            keys = [(f"pts_linears.{l}", W, None) for l in range(D)] + [("views_linears.0", W // 2, None), ("feature_linear", W, None),
                                                                       ("alpha_linear", 1, None), ("rgb_linear", 3, None)]
            blocks = []
            for i, (k, o, _) in enumerate(keys):
                nw = off[2 * i + 1] - off[2 * i]
                blocks.append((k + ".weight", (o, nw // o), off[2 * i]))
                blocks.append((k + ".bias", (o,), off[2 * i + 1]))
            self._blocks, self._nparam = blocks, off[nb]
        return self._blocks

    def flat_params(self, state_dict, prefix="", device="cuda"):
        """state dict -> flat float32 leaf tensor (requires_grad) for an optimizer; missing biases are zeros"""
        blocks = self.param_blocks()
        flat = torch.zeros((self._nparam,), dtype=torch.float32)
        for key, shape, off in blocks:
            if prefix + key in state_dict:
                flat[off:off + int(np.prod(shape))] = torch.as_tensor(_np32(state_dict[prefix + key])).reshape(-1)
        return flat.to(device).requires_grad_(True)

    def unflatten(self, flat):
        return {key: flat[off:off + int(np.prod(shape))].view(shape) for key, shape, off in self.param_blocks()}

    def load_params(self, flat):
        """re-pack every weight stream on the device from new parameter values (after optimizer.step(), run_nerf.py:601)"""
        f = flat.detach().contiguous().float()
        self.param_blocks()
        if f.numel() != self._nparam:
            raise L.EvdError(f"flat parameter tensor has {f.numel()} elements, the network {self._nparam}")
        L.check(L.lib().evd_nerf_load_params(self._h, L.ptr(f), L.stream_ptr()), "evd_nerf_load_params")
        self._synced = (flat.data_ptr(), flat._version, L.backward_generation())

    def mlp_train(self, flat, ray_batch, z_vals, precision=None):
        """NeRF.mlpforward for training: raw [R,S,4] with autograd back to the flat parameter tensor (rays are constants)"""
        if getattr(self, "_synced", None) != (flat.data_ptr(), flat._version, L.backward_generation()):
            self.load_params(flat)
        return _NerfMLP.apply(flat, ray_batch, z_vals, self, precision or self.precision)

    # Backward of mlpforward_train: d_raw [R,S,4] -> flat gradient (what autograd gives for nerf.py:46-72)
    def mlp_backward_flat(self, d_raw, store, precision=None, pts=None, ray_batch=None, accumulate_into=None):
        """flat parameter gradient; with pts [R,S,3] and the ray batch also (flat, d pts [R*S,3], d dirs per sample [R*S,3]), the
        gradients through the two positional encodings.  accumulate_into: a persistent flat float32 gradient buffer the kernels ADD
        into (evd_nerf_grads.accumulate); the returned flat gradient is then None"""
        g = d_raw.contiguous().float()
        R, S = g.shape[:2]
        blocks = self.param_blocks()
        flat = accumulate_into if accumulate_into is not None else torch.zeros((self._nparam,), dtype=torch.float32, device=g.device)
        gs = L.NerfGrads()
        gs.accumulate = int(accumulate_into is not None)
        base = flat.data_ptr()
        for key, shape, off in blocks:
            name, kind = key.rsplit(".", 1)
            field = "_w" if kind == "weight" else "_b"
            if name.startswith("pts_linears."):
                getattr(gs, "pts" + field)[int(name.split(".")[1])] = base + 4 * off
            else:
                head = {"views_linears.0": "views", "feature_linear": "feature", "alpha_linear": "alpha", "rgb_linear": "rgb"}[name]
                setattr(gs, head + field, base + 4 * off)
        nb = int(L.lib().evd_nerf_backward_workspace_bytes())
        ws = torch.empty((nb,), dtype=torch.uint8, device=g.device)
        want = pts is not None
        d_pts = torch.empty((R * S, 3), dtype=torch.float32, device=g.device) if want else None
        d_dirs = torch.empty((R * S, 3), dtype=torch.float32, device=g.device) if want else None
        vd = ray_batch[:, 8:11] if want else None              # a strided view: rows of 11 floats
        L.check(L.lib().evd_nerf_mlp_backward(self._h, L.PREC[precision or self.precision], L.ptr(g), R, S, L.ptr(store), store.numel(),
                                               C.byref(gs), L.ptr(pts), C.c_void_p(vd.data_ptr()) if want else None, 11, L.ptr(d_pts), L.ptr(d_dirs),
                                               L.ptr(ws), nb, L.stream_ptr()), "evd_nerf_mlp_backward")
        if accumulate_into is not None:
            flat = None
        return (flat, d_pts, d_dirs) if want else flat

    def mlp_backward(self, d_raw, store, precision=None):
        """... as {state-dict key: gradient}"""
        return self.unflatten(self.mlp_backward_flat(d_raw, store, precision))

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
