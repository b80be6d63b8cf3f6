"""Mirror of the reference ``networks/renderer.py`` ``NeRFAll`` on libevdnerf.so.

Same method names, argument meaning, return conventions and result-dict keys as the reference
(render :399, render_rays :129, render_path :594, forward :266), so ``run_nerf.py`` can call this object
where it calls ``NeRFAll``. Differences, all forced by moving the random draws and the MLP into kernels:

* random draws are explicit: ``t_rand`` / ``u`` / noise tensors may be passed (keyword-only); when omitted
  and needed they are drawn with torch on the GPU (same distributions as renderer.py:176, rays.py:162,
  nerf.py:99);
* the NaN/Inf guard of render_rays (:259-263, two host syncs per key) is opt-in (``check_numerics=True``);
* ``chunk`` is accepted and honoured, but defaults large: 288 GB of HBM make the reference's memory
  chunking unnecessary.
"""
from __future__ import annotations

import ctypes as C
import time
import types

import numpy as np
import torch

from . import _lib as L
from .nerf import NeRF
from .rays import get_rays


def _args_get(args, name, default):
    return getattr(args, name, default)


class _MergeFeatures(torch.autograd.Function):
    """renderer.py:209-213 + :195 for the training path: out[r, k] = cat([cat([ft0, ftn], 1)[r, order[r, k]], ft_fine[r, k]], -1)
    (evd_merge_features / _bwd; the fine columns are a strided copy)"""

    @staticmethod
    def forward(ctx, ft0, ftn, order, ft_fine, rows=None):
        a, b = ft0.contiguous().float(), ftn.contiguous().float()
        R, S, F = a.shape
        N, Ff = b.shape[1], ft_fine.shape[-1]
        # `rows`: the caller's [R, S + N, F + Ff] buffer whose columns F.. the fine gather has already written (ft_fine is that window)
        placed = rows is not None and rows.shape == (R, S + N, F + Ff) and rows.is_contiguous() and ft_fine.data_ptr() == rows.data_ptr() + 4 * F
        out = rows if placed else torch.empty((R, S + N, F + Ff), dtype=torch.float32, device=a.device)
        L.check(L.lib().evd_merge_features(L.ptr(a), L.ptr(b), L.ptr(order), R, S, N, F, L.ptr(out), F + Ff, L.stream_ptr()), "evd_merge_features")
        if placed:
            ctx.mark_dirty(rows)
        else:
            out[..., F:] = ft_fine.float()
        ctx.save_for_backward(order)
        ctx.dims = (R, S, N, F, Ff)
        return out

    @staticmethod
    def backward(ctx, g):
        (order,) = ctx.saved_tensors
        R, S, N, F, Ff = ctx.dims
        g = g.contiguous().float()
        d0 = torch.empty((R, S, F), dtype=torch.float32, device=g.device)
        dn = torch.empty((R, N, F), dtype=torch.float32, device=g.device)
        L.check(L.lib().evd_merge_features_bwd(L.ptr(g), F + Ff, L.ptr(order), R, S, N, F, L.ptr(d0), L.ptr(dn), L.stream_ptr()), "evd_merge_features_bwd")
        return d0, dn, None, g[..., F:], None          # (the fine columns stay a strided window: the scatter reads them where they are)


def _window(rows, col, width):
    """columns col .. col + width - 1 of a contiguous [R, S, C] float32 buffer as a tensor of its own on the same storage (not an autograd
    view of `rows`: a kernel writes it through its pointer)"""
    R, S, Cc = rows.shape
    return torch.empty(0, dtype=rows.dtype, device=rows.device).set_(rows.untyped_storage(), rows.storage_offset() + col, (R, S, width), (S * Cc, Cc, 1))


class _RayBatch(torch.autograd.Function):
    """NeRFAll.render's ray packing (renderer.py:423-446) under autograd: evd_ray_batch / evd_ray_batch_bwd -- one launch each way instead
    of ~25 small tensor operations and their autograd nodes per render"""

    @staticmethod
    def forward(ctx, rays, H, W, focal, ndc, near, far):
        r = rays.contiguous().float()
        cfg = L.RenderCfg()
        cfg.H, cfg.W, cfg.focal, cfg.ndc, cfg.use_viewdirs, cfg.near, cfg.far = int(H), int(W), float(focal), int(bool(ndc)), 1, float(near), float(far)
        rb = torch.empty((r.shape[0], 11), dtype=torch.float32, device=r.device)
        L.check(L.lib().evd_ray_batch(C.byref(cfg), L.ptr(r), r.shape[0], L.ptr(rb), L.stream_ptr()), "evd_ray_batch")
        ctx.cfg, ctx.rays = cfg, r
        return rb

    @staticmethod
    def backward(ctx, g):
        r = ctx.rays
        d = torch.empty_like(r)
        L.check(L.lib().evd_ray_batch_bwd(C.byref(ctx.cfg), L.ptr(r), L.ptr(g.contiguous().float()), r.shape[0], L.ptr(d), L.stream_ptr()), "evd_ray_batch_bwd")
        return d, None, None, None, None, None, None


class _Points(torch.autograd.Function):
    """pts = rays_o + rays_d z (renderer.py:180,206,235) from the packed ray batch: evd_points / evd_points_bwd (z is a constant: the sample
    positions are detached in the reference)"""

    @staticmethod
    def forward(ctx, rb, z):
        if rb.shape[1] != 11:
            raise L.EvdError("points under autograd needs the 11-column ray batch")
        R, S = z.shape
        pts = torch.empty((R, S, 3), dtype=torch.float32, device=rb.device)
        L.check(L.lib().evd_points(L.ptr(rb), rb.shape[1], L.ptr(z), R, S, L.ptr(pts), L.stream_ptr()), "evd_points")
        ctx.z, ctx.ncol = z, rb.shape[1]
        return pts

    @staticmethod
    def backward(ctx, g):
        z = ctx.z
        R, S = z.shape
        d_rb = torch.empty((R, 11), dtype=torch.float32, device=z.device)       # accumulate = 0: the kernel writes whole rows
        L.check(L.lib().evd_points_bwd(L.ptr(z), L.ptr(g.contiguous().float()), R, S, 0, L.ptr(d_rb), L.stream_ptr()), "evd_points_bwd")
        return d_rb, None


def points(rb, z):
    """[R,S,3] sample positions of a packed ray batch [R,11] at depths z [R,S]; differentiable w.r.t. the batch when it requires grad"""
    rb, z = rb.contiguous().float(), z.contiguous().float()
    if rb.requires_grad and torch.is_grad_enabled():
        return _Points.apply(rb, z)
    R, S = z.shape
    pts = torch.empty((R, S, 3), dtype=torch.float32, device=rb.device)
    L.check(L.lib().evd_points(L.ptr(rb), rb.shape[1], L.ptr(z), R, S, L.ptr(pts), L.stream_ptr()), "evd_points")
    return pts


class _FlatParams(torch.autograd.Function):
    """A level's flat parameter tensor as a function of its per-tensor leaves.  The leaves ARE views of the flat tensor's storage
    (enable_training), so the forward is an alias, not a torch.cat; the backward hands every leaf its slice of the flat gradient (views,
    no kernels).  In the opt-in in-place mode the consuming node adds its gradient straight into the level's persistent flat gradient
    buffer (whose slices are the leaves' .grad) and this node receives None."""

    @staticmethod
    def forward(ctx, flat, level, *leaves):
        ctx.level = level
        ctx.set_materialize_grads(False)
        return flat.detach()

    @staticmethod
    def backward(ctx, g):
        lv = ctx.level
        if g is None:
            return (None, None) + (None,) * len(lv.blocks)
        return (None, None) + tuple(g[o:o + n].view(shape) for (_, shape, o), n in zip(lv.blocks, lv.sizes))


class _Level:
    """Trainable state of one network / PDRF level: `flat` is the storage of its sigma / colour (or NeRF MLP) parameters in the library's
    canonical order, `leaves` one autograd leaf per reference parameter, each a VIEW of `flat` (an optimizer's in-place update of a
    leaf updates the storage the library re-packs from, and bumps its shared version counter); `grids` the tri-plane leaves."""

    def __init__(self, net, prefix, flat, blocks, grids, in_place, absent=()):
        """absent: parameter blocks of the library's layout that the reference model does NOT have (the colour networks' biases with
        rgb_add_bias off, voxnerf.py:60,80 / nerf.py:37 -- every shipped config): they stay zero in `flat`, get no leaf, appear in no
        optimizer group or state dict, and whatever the backward kernels write into their gradient slices is never read."""
        from collections import OrderedDict
        self.all_blocks = blocks
        blocks = [b for b in blocks if b[0] not in absent]
        self.net, self.prefix, self.flat, self.blocks, self.grids, self.in_place = net, prefix, flat, blocks, grids, in_place
        self.sizes = [int(np.prod(shape)) for _, shape, _ in blocks]
        self.leaves = OrderedDict((prefix + key, flat[o:o + n].view(shape).detach().requires_grad_(True)) for (key, shape, o), n in zip(blocks, self.sizes))
        self.flat_grad = None

    def params(self):
        flat = _FlatParams.apply(self.flat, self, *self.leaves.values())
        flat._evd_accum = self.attach_grads if self.in_place else None       # read by the consuming node (_VoxelMLP / _NerfMLP)
        return {"net": flat, "grids": list(self.grids.values())} if self.grids is not None else flat

    def attach_grads(self):
        """in-place mode: the flat gradient buffer whose slices are the leaves' .grad.  When the optimizer dropped the gradients
        (zero_grad(set_to_none=True), the default) the buffer is zeroed by ONE fill and re-attached."""
        none = [t.grad is None for t in self.leaves.values()]
        if self.flat_grad is None or all(none):
            if self.flat_grad is None:
                self.flat_grad = torch.zeros_like(self.flat)
            else:
                self.flat_grad.zero_()
            for t, (_, shape, o), n in zip(self.leaves.values(), self.blocks, self.sizes):
                t.grad = self.flat_grad[o:o + n].view(shape)
        elif any(none):
            raise L.EvdError(f"{self.prefix}: in-place gradient accumulation needs the gradients of ALL of a level's parameters dropped or kept "
                             "together (optimizer.zero_grad() / model.zero_grad()); some are None")
        return self.flat_grad


class NeRFAll:
    """mode='nerf' (two NeRF MLPs, renderer.py:82-100) or mode='c2f' (PDRF coarse/fine levels, :46-81)."""

    def __init__(self, args, state_dict, kernelsnet=None, awpnet=None, precision="f16x3", device=None):
        self.args = args
        self.mode = args.mode
        if self.mode not in ("nerf", "c2f"):
            raise NotImplementedError(f"{self.mode} for rendering network is not implemented")
        self.kernel_type = _args_get(args, "kernel_type", None)
        self.kernelsnet = kernelsnet
        self.awpnet = awpnet
        self.use_awp = bool(_args_get(args, "kernel_use_awp", False)) and awpnet is not None
        self.extract_feature = "before_linear" if self.use_awp else "after_linear"
        # precision: the arithmetic of the MLP GEMMs (include/evdnerf.h EVD_PREC_*).  "f16m" is a TRAINING mode (split-float16 forward, float16
        # backward); a model built with it renders its inference passes in "f16x3", the same forward arithmetic
        self.train_precision = precision
        precision = "f16x3" if precision == "f16m" else precision
        self.precision = precision
        self.training = False
        self.device = torch.device(device or "cuda")
        self.use_viewdirs = bool(args.use_viewdirs)
        if not self.use_viewdirs and self.mode != "nerf":
            raise NotImplementedError("use_viewdirs=False is built for mode='nerf' (the PDRF levels always take view directions, voxnerf.py:248)")
        self.mlp_fine = None
        if self.mode == "nerf":
            common = dict(use_viewdirs=self.use_viewdirs, D=args.netdepth, W=args.netwidth, multires=args.multires, multires_views=args.multires_views,
                          rgb_activate=args.rgb_activate, sigma_activate=args.sigma_activate,
                          render_rmnearplane=_args_get(args, "render_rmnearplane", 0),
                          extract_feature=self.extract_feature, composite_feature=False, precision=precision)
            # kernel_type PBE: the coarse network's feature map is composited (renderer.py:30-34,89; nerf.py:167-169)
            self.mlp_coarse = NeRF(state_dict, "mlp_coarse.", **dict(common, composite_feature=self.kernel_type == "PBE"))
            if args.N_importance > 0:
                common_f = dict(common, D=_args_get(args, "netdepth_fine", args.netdepth), W=_args_get(args, "netwidth_fine", args.netwidth))
                self.mlp_fine = NeRF(state_dict, "mlp_fine.", **common_f)
        else:
            from .voxnerf import VoxelNeRFRayFeatures, VoxelNeRFSampleFeatures
            # kernel_type PBE: the coarse level composites its geo features and runs its colour network per ray (renderer.py:30-34,
            # voxnerf.py:223-239) -- built for inference (render / coarse_render); its training forward needs the PDRF blur model
            # (networks/pdrf/blurmodel.py, out of scope: no shipped config uses it)
            ic = 3 * (1 + 2 * args.multires)
            rm = _args_get(args, "render_rmnearplane", 0)
            self.mlp_coarse = VoxelNeRFRayFeatures(
                state_dict, "mlp_coarse.", args.bounding_box, num_layers=args.coarse_num_layers, hidden_dim=args.coarse_hidden_dim,
                geo_feat_dim=_args_get(args, "kernel_feat_cnl", 15), num_layers_color=args.coarse_num_layers_color,
                input_ch=args.coarse_app_dim + ic, multires=args.multires, multires_views=args.multires_views,
                render_rmnearplane=rm, app_dim=args.coarse_app_dim, app_n_comp=args.coarse_app_n_comp, n_voxels=args.coarse_n_voxels,
                app_actfn=_args_get(args, "coarse_app_actfn", "none"), composite_feature=self.kernel_type == "PBE", precision=precision)
            if args.N_importance > 0:
                self.mlp_fine = VoxelNeRFSampleFeatures(
                    state_dict, "mlp_fine.", args.bounding_box, num_layers=args.fine_num_layers, hidden_dim=args.fine_hidden_dim,
                    geo_feat_dim=args.fine_geo_feat_dim, num_layers_color=args.fine_num_layers_color,
                    input_ch=args.coarse_app_dim + args.fine_app_dim + ic, multires=args.multires, multires_views=args.multires_views,
                    render_rmnearplane=rm, app_dim=args.fine_app_dim, app_n_comp=args.fine_app_n_comp, n_voxels=args.fine_n_voxels,
                    app_actfn=_args_get(args, "fine_app_actfn", "none"), precision=precision)
        self._ws = None

    # nn.Module-like switches used by run_nerf.py
    def sync_parameters(self):
        """push the current values of the trainable tensors into the library (re-pack the weight streams, reload the grids); the
        training forward does this lazily, an evaluation render after optimizer.step() needs it explicitly (train(False) calls it)"""
        if getattr(self, "_levels", None) is None:
            return
        with torch.no_grad():
            tp = self._current_params()
        for net, p in ((self.mlp_coarse, tp[0]), (self.mlp_fine, tp[1])):
            if p is None:
                continue
            if isinstance(p, dict):
                net.load_params(p["net"])
                net._sync(p["grids"])
            else:
                net.load_params(p)

    def invalidate_packed(self):
        """Mark the library's packed copies of the trainable tensors stale: the next training forward (or sync_parameters / eval) re-packs the
        weight streams and reloads the grids.  The re-pack check sees a changed tensor address or version counter and any backward of the
        library's nodes -- NOT a write that bumps no version and follows no backward: `p.data.copy_(...)` (a checkpoint or EMA restore),
        a second fused optimizer step, a manual reload.  Call this after such a write (load_state_dict-style helpers of a training loop)."""
        for net in (self.mlp_coarse, self.mlp_fine):
            if net is not None:
                net._synced = net._synced_net = None

    def train(self, mode=True):
        if not mode:
            self.sync_parameters()
        self.training = mode
        self.mlp_coarse.train(mode)
        if self.mlp_fine is not None:
            self.mlp_fine.train(mode)
        return self

    def eval(self):
        return self.train(False)

    def _cfg(self, H, W, focal, ndc, near, far, N_samples, N_importance, lindisp, perturb, white_bkgd):
        c = L.RenderCfg()
        c.H, c.W, c.focal = int(H), int(W), float(focal)
        c.ndc, c.use_viewdirs, c.lindisp = int(bool(ndc)), int(getattr(self, "use_viewdirs", True)), int(bool(lindisp))
        c.N_samples, c.N_importance, c.white_bkgd = int(N_samples), int(N_importance), int(bool(white_bkgd))
        c.near, c.far, c.perturb = float(near), float(far), float(perturb)
        c.is_train, c.precision = int(self.training), L.PREC[self.precision]
        return c

    def _workspace(self, cfg, R):
        if self.mode == "c2f":
            need = int(L.lib().evd_c2f_render_workspace_bytes(self.mlp_coarse.handle, self.mlp_fine.handle if self.mlp_fine else None,
                                                              C.byref(cfg), R))
        else:
            need = int(L.lib().evd_nerf_render_workspace_bytes(C.byref(cfg), R))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty((need,), dtype=torch.uint8, device=self.device)
        return self._ws, need

    # ------------------------------------------------------------------ render_rays, renderer.py:129-264
    def render_rays(self, ray_batch, N_samples, retraw=False, lindisp=False, perturb=0., N_importance=0,
                    white_bkgd=False, raw_noise_std=0., pytest=False, force_naive=False, inference=False, *,
                    t_rand=None, u=None, noise0=None, noise1=None, check_numerics=False, _cfg=None, _rays=None):
        dev = self.device
        if _rays is None:
            rb = ray_batch.contiguous().float()
            R = rb.shape[0]
            if rb.shape[1] != (11 if self.use_viewdirs else 8):
                raise L.EvdError("render_rays needs the 11-column ray batch (o, d, near, far, viewdirs); 8 columns with use_viewdirs=False")
        else:
            rb, R = None, _rays.shape[0]
        cfg = _cfg or self._cfg(0, 0, 1.0, False, 0., 1., N_samples, N_importance, lindisp, perturb, white_bkgd)
        cfg.N_samples, cfg.N_importance, cfg.lindisp = int(N_samples), int(N_importance), int(bool(lindisp))
        cfg.perturb, cfg.white_bkgd, cfg.is_train = float(perturb), int(bool(white_bkgd)), int(self.training)
        S, Ni = int(N_samples), int(N_importance)
        St = S + Ni
        f32 = dict(dtype=torch.float32, device=dev)
        if perturb > 0.:
            if t_rand is None:
                t_rand = torch.rand((R, S), **f32)
            if Ni > 0 and u is None:
                u = torch.rand((R, Ni), **f32)
        if raw_noise_std > 0.:
            if noise0 is None:
                noise0 = torch.randn((R, S - 1), **f32) * raw_noise_std
            if Ni > 0 and noise1 is None:
                noise1 = torch.randn((R, St - 1), **f32) * raw_noise_std
        want_feat = self.use_awp and not force_naive and not inference
        ret = {"rgb_map": torch.empty((R, 3), **f32), "depth_map": torch.empty((R,), **f32),
               "acc_map": torch.empty((R,), **f32)}
        out = L.RenderOut()
        out.rgb, out.depth, out.acc = L.ptr(ret["rgb_map"]), L.ptr(ret["depth_map"]), L.ptr(ret["acc_map"])
        if retraw or want_feat:
            ret["z_vals"] = torch.empty((R, St), **f32)
            out.z_vals = L.ptr(ret["z_vals"])
        if retraw:
            ret["weights"] = torch.empty((R, St), **f32)
            out.weights = L.ptr(ret["weights"])
        if Ni > 0:
            ret["rgb0"], ret["depth0"], ret["acc0"] = torch.empty((R, 3), **f32), torch.empty((R,), **f32), torch.empty((R,), **f32)
            ret["z_std"] = torch.empty((R,), **f32)
            out.rgb0, out.depth0, out.acc0, out.z_std = (L.ptr(ret["rgb0"]), L.ptr(ret["depth0"]), L.ptr(ret["acc0"]),
                                                         L.ptr(ret["z_std"]))
            if retraw:
                ret["z_vals0"], ret["weights0"] = torch.empty((R, S), **f32), torch.empty((R, S), **f32)
                out.z_vals0, out.weights0 = L.ptr(ret["z_vals0"]), L.ptr(ret["weights0"])
        if want_feat:
            last = self.mlp_fine if (Ni > 0 and self.mlp_fine is not None) else self.mlp_coarse
            if self.mode != "nerf" and getattr(last, "composite_feature", False):
                ret["depth_feature"] = torch.empty((R, last.geo_feat_dim), **f32)          # the composited map of a PBE level (voxnerf.py:226)
            else:
                ret["depth_feature"] = torch.empty((R, St, last.W if self.mode == "nerf" else last.geo_feat_dim), **f32)
            out.feature = L.ptr(ret["depth_feature"])
        out.feature_kind = 2 if self.extract_feature == "before_linear" else 1
        ws, need = self._workspace(cfg, R)
        fine = self.mlp_fine.handle if (Ni > 0 and self.mlp_fine is not None) else None
        tr = t_rand.contiguous().float() if t_rand is not None else None
        uu = u.contiguous().float() if u is not None else None
        n0 = noise0.contiguous().float() if noise0 is not None else None
        n1 = noise1.contiguous().float() if noise1 is not None else None
        fn_rays, fn_render = ((L.lib().evd_nerf_render_rays, L.lib().evd_nerf_render) if self.mode == "nerf" else
                              (L.lib().evd_c2f_render_rays, L.lib().evd_c2f_render))
        if _rays is None:
            L.check(fn_rays(self.mlp_coarse.handle, fine, C.byref(cfg), L.ptr(rb), R, L.ptr(tr), L.ptr(uu),
                                                 L.ptr(n0), L.ptr(n1), C.byref(out), L.ptr(ws), need, L.stream_ptr()),
                    "render_rays")
        else:
            L.check(fn_render(self.mlp_coarse.handle, fine, C.byref(cfg), L.ptr(_rays), R, L.ptr(tr), L.ptr(uu),
                                            L.ptr(n0), L.ptr(n1), C.byref(out), L.ptr(ws), need, L.stream_ptr()),
                    "render")
        if not retraw and "z_vals" in ret and not want_feat:
            del ret["z_vals"]
        if check_numerics:
            # renderer.py:259-263 as ONE launch over all keys into a device flag word per key (evd_numerics_flags).
            # check_numerics="device": no synchronisation, the int32 tensor is returned as ret["numerics_flags"] (key order =
            # ret["numerics_keys"]); check_numerics=True: read it back (one sync instead of two per key) and print the
            # reference's messages
            keys = [k for k in ret if ret[k].dtype == torch.float32]
            flags = self.numerics_flags([ret[k] for k in keys])
            if check_numerics == "device":
                ret["numerics_flags"], ret["numerics_keys"] = flags, keys
            else:
                for k, f in zip(keys, flags.tolist()):
                    if f & 1:
                        print(f"! [Numerical Error] {k} contains nan.")
                    if f & 2:
                        print(f"! [Numerical Error] {k} contains inf.")
        return ret

    def numerics_flags(self, tensors):
        """int32 [len(tensors)] on the device: bit 0 = NaN present, bit 1 = Inf present (no host synchronisation)"""
        ts = [t.contiguous() for t in tensors]
        n = len(ts)
        flags = torch.empty((n,), dtype=torch.int32, device=self.device)
        ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in ts])
        counts = (C.c_long * n)(*[t.numel() for t in ts])
        L.check(L.lib().evd_numerics_flags(ptrs, counts, n, L.ptr(flags), L.stream_ptr()), "evd_numerics_flags")
        return flags

    # ------------------------------------------------------------------ render_rays under autograd (mode='nerf')
    def trainable_parameters(self, state_dict):
        """mode='nerf': (flat_coarse, flat_fine | None), one float32 leaf tensor per network (NeRF.flat_params).
        mode='c2f': ({"net": flat, "grids": [plane0..2, line0..2, basis]}, same for the fine level | None) -- the level's
        sigma/colour parameters as one flat tensor and its tri-plane parameters in the library's channel-last layout."""
        if self.mode == "c2f":
            def level(m, prefix):
                return {"net": m.flat_params(state_dict, prefix, self.device), "grids": m.grid_params()}
            return level(self.mlp_coarse, "mlp_coarse."), (level(self.mlp_fine, "mlp_fine.") if self.mlp_fine is not None else None)
        fc = self.mlp_coarse.flat_params(state_dict, "mlp_coarse.", self.device)
        ff = self.mlp_fine.flat_params(state_dict, "mlp_fine.", self.device) if self.mlp_fine is not None else None
        return fc, ff

    def render_rays_train(self, ray_batch, flat_coarse, flat_fine, N_samples, N_importance=0, lindisp=False, perturb=0.,
                          white_bkgd=False, raw_noise_std=0., *, t_rand=None, u=None, noise0=None, noise1=None, want_feature=False):
        """renderer.py:129-264 (else-branch) with gradients to the two flat parameter tensors: stratified z, fused MLP
        (forward keeps activations, hand-written backward), compositing scan (autograd node), hierarchical resampling on the
        detached coarse weights (renderer.py:233 z_samples.detach()), fine pass.  Rays are constants (no pose gradients)."""
        from .rays import sample_pdf_merge
        rb = ray_batch.contiguous().float()
        R, S, Ni = rb.shape[0], int(N_samples), int(N_importance)
        f32 = dict(dtype=torch.float32, device=rb.device)
        cfg = self._cfg(0, 0, 1.0, False, 0., 1., S, Ni, lindisp, perturb, white_bkgd)
        cfg.is_train = 1
        if perturb > 0.:
            if t_rand is None and u is None and Ni > 0:           # one draw for both (renderer.py:171 t_rand, utils/rays.py:166 u)
                rnd = torch.rand((R * (S + Ni),), **f32)
                t_rand, u = rnd[:R * S].view(R, S), rnd[R * S:].view(R, Ni)
            t_rand = torch.rand((R, S), **f32) if t_rand is None else t_rand
            u = torch.rand((R, Ni), **f32) if (Ni > 0 and u is None) else u
        if raw_noise_std > 0.:
            noise0 = torch.randn((R, S - 1), **f32) * raw_noise_std if noise0 is None else noise0
            noise1 = torch.randn((R, S + Ni - 1), **f32) * raw_noise_std if (Ni > 0 and noise1 is None) else noise1
        z0 = torch.empty((R, S), **f32)
        tr = t_rand.contiguous().float() if t_rand is not None else None
        L.check(L.lib().evd_sample_z(C.byref(cfg), L.ptr(rb), 11, R, L.ptr(tr), L.ptr(z0), L.stream_ptr()), "evd_sample_z")
        if self.mode == "c2f":
            return self._render_rays_train_c2f(rb, z0, flat_coarse, flat_fine, S, Ni, perturb, u, noise0, noise1, want_feature)
        rays_d = rb[:, 3:6].contiguous()
        if want_feature:
            raise NotImplementedError("depth_feature under autograd is built for mode='c2f' (the shipped AWP configs)")
        raw0 = self.mlp_coarse.mlp_train(flat_coarse, rb, z0, self.train_precision)
        rgb0, _, acc0, w0, depth0, _ = self.mlp_coarse.raw2outputs(raw0, z0, rays_d, None, 0., white_bkgd, noise=noise0)
        if Ni <= 0:
            return {"rgb_map": rgb0, "depth_map": depth0, "acc_map": acc0, "weights": w0, "z_vals": z0}
        _, zm, _, zstd = sample_pdf_merge(z0, w0.detach(), Ni, det=(perturb == 0.), u=u)
        raw1 = self.mlp_fine.mlp_train(flat_fine, rb, zm, self.train_precision)
        rgb, _, acc, w1, depth, _ = self.mlp_fine.raw2outputs(raw1, zm, rays_d, None, 0., white_bkgd, noise=noise1)
        return {"rgb_map": rgb, "depth_map": depth, "acc_map": acc, "weights": w1, "z_vals": zm, "rgb0": rgb0, "depth0": depth0,
                "acc0": acc0, "z_std": zstd}

    def _render_rays_train_c2f(self, rb, z0, pc, pf, S, Ni, perturb, u, noise0, noise1, want_feature=False):
        """renderer.py:182-217 under autograd: tri-plane gathers (scatter-add backward), level networks (fused forward/backward),
        compositing scans, resampling on the detached coarse weights; coarse features of the merged set are the re-ordered rows
        of the old and new points (:209-213), fine features are sampled at the merged points."""
        from .rays import sample_pdf_merge
        coarse, fine = self.mlp_coarse, self.mlp_fine
        vd = rb[:, 8:11].contiguous()
        rays_d = rb[:, 3:6].contiguous()
        pts0 = points(rb, z0)
        ft0 = coarse.sample_train(pts0, pc["grids"], self.train_precision)
        raw0 = coarse.mlp_train(pc["net"], pts0, vd, ft0, self.train_precision)
        rgb0, _, acc0, w0, depth0 = coarse.raw2outputs(raw0, z0, rays_d, is_train=True, noise=noise0)
        if Ni <= 0:
            return {"rgb_map": rgb0, "depth_map": depth0, "acc_map": acc0, "weights": w0, "z_vals": z0}
        zs, zm, order, zstd = sample_pdf_merge(z0, w0.detach(), Ni, det=(perturb == 0.), u=u, want_order=True)
        ftn = coarse.sample_train(points(rb, zs), pc["grids"], self.train_precision)
        ptm = points(rb, zm)
        # cat([coarse features re-ordered by the sort (:209-213), fine features at the merged points]) in one row buffer: the merge is a
        # library kernel writing columns 0..fc-1 (a row permutation: its backward is one too), the fine features land behind them
        # -- and the fine gather writes them there itself (its output is a strided window of the row buffer; no copy either way)
        fc, ff = ft0.shape[-1], fine.app_dim
        rows = torch.empty((ptm.shape[0], ptm.shape[1], fc + ff), dtype=torch.float32, device=ptm.device)
        ft = _MergeFeatures.apply(ft0, ftn, order, fine.sample_train(ptm, pf["grids"], self.train_precision, out=_window(rows, fc, ff)), rows)
        feat = None
        if want_feature == "fragments":                     # fused AWP consumer: the geo features stay in the level's store (awp.FusedAWP)
            from .voxnerf import GeoFragments
            feat = GeoFragments()
            raw1, feat.token = fine.mlp_train(pf["net"], ptm, vd, ft, self.train_precision, want_feature=feat)
        elif want_feature:
            raw1, feat = fine.mlp_train(pf["net"], ptm, vd, ft, self.train_precision, want_feature=True)
        else:
            raw1 = fine.mlp_train(pf["net"], ptm, vd, ft, self.train_precision)
        rgb, _, acc, w1, depth = fine.raw2outputs(raw1, zm, rays_d, is_train=True, noise=noise1)
        ret = {"rgb_map": rgb, "depth_map": depth, "acc_map": acc, "weights": w1, "z_vals": zm, "rgb0": rgb0, "depth0": depth0,
               "acc0": acc0, "z_std": zstd}
        if feat is not None:
            ret["depth_feature"] = feat                    # the fine level's per-sample geo features (renderer.py:253-256), for AWP
        return ret

    @staticmethod
    def ray_batch_train(H, W, K, rays, ndc=True, near=0., far=1.):
        """NeRFAll.render's ray packing (renderer.py:423-446: viewdirs = d / |d| BEFORE the NDC warp of utils/rays.py:104-145 with
        near plane 1): rays [R,3,2] -> ray batch [R,11], differentiable w.r.t. the rays -- this is what lets the loss reach the blur
        kernel's warped rays.  On the GPU: evd_ray_batch and its hand-written backward (one launch each); elsewhere (CPU tests of
        the formula) the same arithmetic in torch."""
        if rays.is_cuda:
            if rays.requires_grad and torch.is_grad_enabled():
                return _RayBatch.apply(rays, H, W, float(K[0][0]), ndc, near, far)
            with torch.no_grad():
                return _RayBatch.apply(rays, H, W, float(K[0][0]), ndc, near, far)
        o, d = rays[..., 0], rays[..., 1]
        vd = d / torch.linalg.norm(d, dim=-1, keepdim=True)
        if ndc:
            focal = float(K[0][0])
            cw, ch = -1.0 / (W / (2.0 * focal)), -1.0 / (H / (2.0 * focal))
            t = -(1.0 + o[..., 2]) / d[..., 2]
            o = o + t[..., None] * d
            ox, oy, o2 = o[..., 0] / o[..., 2], o[..., 1] / o[..., 2], 1.0 + 2.0 / o[..., 2]
            d = torch.stack([cw * (d[..., 0] / d[..., 2] - ox), ch * (d[..., 1] / d[..., 2] - oy), 1.0 - o2], -1)
            o = torch.stack([cw * ox, ch * oy, o2], -1)
        nf = torch.tensor([near, far], dtype=o.dtype, device=o.device).expand(o.shape[0], 2)
        return torch.cat([o, d, nf, vd], -1)

    def forward_train(self, H, W, K, rays, params_coarse, params_fine, rays_info=None, force_naive=True, ndc=True, near=0., far=1.,
                      N_samples=64, N_importance=0, tv=True, **kw):
        """The training branch of NeRFAll.forward (renderer.py:266-397) under autograd: [blur kernel -> P warped rays per pixel]
        -> ray packing -> render_rays_train -> [composition with the kernel's weights] ; returns the reference's
        (rgb, rgb0, other_loss, other_tensors).  Gradients reach params_coarse / params_fine (trainable_parameters) and, through
        the rays, the kernelsnet (a PyTorch module, as in the reference).  kw: lindisp, perturb, white_bkgd, raw_noise_std and the
        explicit random draws of render_rays_train."""
        other_loss, other_tensors = {}, {}
        use_kernel = self.kernelsnet is not None and not force_naive
        if use_kernel:
            if self.kernel_type != "RBK":
                raise NotImplementedError("only the RBK kernel of the shipped configs is supported")
            new_rays, weight1, align_loss, extra1 = self.kernelsnet(H, W, K, rays, rays_info, feats=None, return_img_embed=self.use_awp)
            ray_num, pt_num = new_rays.shape[:2]
            flat_rays = new_rays.reshape(-1, 3, 2)
        else:
            flat_rays = rays
        rb = self.ray_batch_train(H, W, K, flat_rays, ndc, near, far)
        awp = use_kernel and self.use_awp
        from .awp import FusedAWP
        # (the fragment coupling is a half-precision one: in the float32-grade mode the fused module takes the feature rows)
        # (... in the float16 mode's fragment format, which the mixed training modes f16c / f16m keep too)
        store_prec = "f16" if self.train_precision in ("f16c", "f16m") else self.train_precision
        fused = awp and isinstance(self.awpnet, FusedAWP) and self.mode == "c2f" and store_prec in ("f16", "bf16") and \
            self.awpnet.embed.precision == store_prec
        out = self.render_rays_train(rb, params_coarse, params_fine, N_samples, N_importance, want_feature="fragments" if fused else awp, **kw)
        rgb, rgb0 = out["rgb_map"], out.get("rgb0")
        if awp:         # adaptive weight proposal on the fine level's per-sample features (renderer.py:310-316): a second composition
            ccw = self.awpnet(out["depth_feature"], out["z_vals"], rb[:, 3:6], extra1["img_embed"])
            ccw = ccw + ccw * self.awpnet.ccw_fine_scale
            ccw = ccw / torch.sum(ccw, -1, keepdim=True)
            other_tensors["rgb_awp"] = (rgb.reshape(ray_num, pt_num, 3) * ccw[..., None]).sum(1)
        if use_kernel:
            rgb_pts = rgb.reshape(ray_num, pt_num, 3)
            rgb = (rgb_pts * weight1[..., None]).sum(1)                         # rbk_weighted_sum, blurmodel.py:112-127
            other_tensors["stage1_rgb_pts0"] = rgb_pts[:, 0]
            if rgb0 is not None:
                rgb0_pts = rgb0.reshape(ray_num, pt_num, 3)
                rgb0 = (rgb0_pts * weight1[..., None]).sum(1)
                other_tensors["stage1_rgb1_pts0"] = rgb0_pts[:, 0]
            if align_loss is not None:
                other_loss["align"] = align_loss.reshape(1, 1)
            other_tensors.update({f"stage1_{k}": v for k, v in extra1.items()})
        else:
            other_tensors["stage1_rgb_pts0"] = rgb
            if rgb0 is not None:
                other_tensors["stage1_rgb1_pts0"] = rgb0
        if tv and self.mode == "c2f":
            other_loss["TV"] = self.tv_loss_train(params_coarse, params_fine if N_importance > 0 else None)
        return rgb, rgb0, other_loss, other_tensors

    def tv_loss_train(self, pc, pf=None):
        """other_loss['TV'] of the training forward (renderer.py:361-365) with gradients to the grids"""
        tv = self.mlp_coarse.tv_loss_train(pc["grids"])
        if pf is not None and self.mlp_fine is not None:
            tv = tv + self.mlp_fine.tv_loss_train(pf["grids"])
        return tv * 5

    # ------------------------------------------------------------------ render, renderer.py:399-466
    def render(self, H, W, K, chunk=1 << 22, rays=None, c2w=None, ndc=True, near=0., far=1., use_viewdirs=False,
               c2w_staticcam=None, **kwargs):
        """renderer.py:399-466.  `rays` [..., 3, 2]; as in the reference, `c2w` is not read when rays are given (render_path passes
        both, :614) -- without rays the full H x W image of `c2w` is generated on the device (evd_get_rays).  `c2w_staticcam`
        (:427-430): origins / directions of that camera, view directions of the given rays.  use_viewdirs must match the model
        (args.use_viewdirs): False packs the 8-column batch of :443-446 for networks with the output_linear head (mode='nerf' only)."""
        if bool(use_viewdirs) != self.use_viewdirs:
            raise L.EvdError(f"render(use_viewdirs={use_viewdirs}) on a model built with use_viewdirs={self.use_viewdirs} (renderer.py:443-446: "
                             "the ray batch has 11 columns with view directions, 8 without)")
        focal = float(K[0][0])
        if rays is None:
            if c2w is None:
                raise L.EvdError("render needs rays or c2w")
            o, d = get_rays(H, W, K, torch.as_tensor(c2w, device=self.device))
            rays = torch.stack([o, d], dim=-1)
        rays = rays.to(self.device).float()
        sh = rays.shape[:-2]                      # [..., 3, 2]
        flat = rays.reshape(-1, 3, 2).contiguous()
        R = flat.shape[0]
        N_samples = kwargs.get("N_samples")
        cfg = self._cfg(H, W, focal, ndc, near, far, N_samples, kwargs.get("N_importance", 0), kwargs.get("lindisp", False),
                        kwargs.get("perturb", 0.), kwargs.get("white_bkgd", False))
        batch = None
        if c2w_staticcam is not None and self.use_viewdirs:          # (read inside `if use_viewdirs` only, renderer.py:424-430)
            so, sd_ = get_rays(H, W, K, torch.as_tensor(c2w_staticcam, device=self.device))
            cam = torch.stack([so, sd_], dim=-1).reshape(-1, 3, 2).contiguous().float()
            if cam.shape[0] != R:
                raise L.EvdError(f"c2w_staticcam gives {cam.shape[0]} rays, the batch has {R}")
            batch = torch.empty((R, 11), dtype=torch.float32, device=self.device)
            L.check(L.lib().evd_ray_batch(C.byref(cfg), L.ptr(cam), R, L.ptr(batch), L.stream_ptr()), "evd_ray_batch")
            vd = flat[..., 1]
            batch[:, 8:11] = vd / torch.norm(vd, dim=-1, keepdim=True)
        all_ret = {}
        numerics = None
        rand_keys = ("t_rand", "u", "noise0", "noise1")
        for i in range(0, max(1, R), chunk):
            kw = dict(kwargs)
            for k in rand_keys:
                if kw.get(k) is not None:
                    kw[k] = kw[k][i:i + chunk]
            if batch is not None:
                ret = self.render_rays(batch[i:i + chunk], _cfg=cfg, **kw)
            else:
                ret = self.render_rays(None, _cfg=cfg, _rays=flat[i:i + chunk], **kw)
            if "numerics_flags" in ret:                  # check_numerics="device": OR the chunks' flag words, stay on the device
                nf, nk = ret.pop("numerics_flags"), ret.pop("numerics_keys")
                numerics = ((nf if numerics is None else numerics[0] | nf), nk)
            for k, v in ret.items():
                all_ret.setdefault(k, []).append(v)
        all_ret = {k: (v[0] if len(v) == 1 else torch.cat(v, 0)) for k, v in all_ret.items()}
        for k in all_ret:
            all_ret[k] = all_ret[k].reshape(list(sh) + list(all_ret[k].shape[1:]))
        k_extract = ["rgb_map", "depth_map", "acc_map"]
        ret_list = [all_ret[k] for k in k_extract]
        ret_dict = {k: all_ret[k] for k in all_ret if k not in k_extract}
        if numerics is not None:
            ret_dict["numerics_flags"], ret_dict["numerics_keys"] = numerics
        if self.use_awp:
            # NDC ray directions, renderer.py:464-465
            from .rays import get_ndc_rays
            rd = flat[..., 1]
            if ndc:
                _, rd = get_ndc_rays(H, W, focal, 1., flat[..., 0].contiguous(), rd.contiguous())
            ret_dict["rays_d"] = rd.reshape(-1, 3)
        return ret_list + [ret_dict]

    # ------------------------------------------------------------------ coarse_render, renderer.py:468-592
    def coarse_render_rays(self, ray_batch, N_samples, retraw=False, lindisp=False, perturb=0., N_importance=0, white_bkgd=False,
                           raw_noise_std=0., force_naive=False, inference=False, *, t_rand=None, noise0=None, _cfg=None, _rays=None):
        """renderer.py:525-592: the COARSE level alone on N_samples stratified samples -> (rgb_map [R,3], feat); feat is what the coarse
        backbone returns as its feature map: per-sample geo features [R,S,geo] (mode='c2f'), their composite [R,geo] for a PBE level
        (voxnerf.py:226), the per-sample feature of NeRF.forward (mode='nerf').  N_importance is accepted and ignored, as there."""
        dev = self.device
        R = ray_batch.shape[0] if _rays is None else _rays.shape[0]
        S = int(N_samples)
        cfg = _cfg or self._cfg(0, 0, 1.0, False, 0., 1., S, 0, lindisp, perturb, white_bkgd)
        cfg.N_samples, cfg.N_importance, cfg.lindisp = S, 0, int(bool(lindisp))
        cfg.perturb, cfg.white_bkgd, cfg.is_train = float(perturb), int(bool(white_bkgd)), int(self.training)
        f32 = dict(dtype=torch.float32, device=dev)
        if perturb > 0. and t_rand is None:
            t_rand = torch.rand((R, S), **f32)
        if raw_noise_std > 0. and noise0 is None:
            noise0 = torch.randn((R, S - 1), **f32) * raw_noise_std
        c = self.mlp_coarse
        rgb, depth, acc = torch.empty((R, 3), **f32), torch.empty((R,), **f32), torch.empty((R,), **f32)
        if self.mode == "nerf":
            feat = torch.empty((R, S, c.W), **f32)
        else:
            feat = torch.empty((R, c.geo_feat_dim) if c.composite_feature else (R, S, c.geo_feat_dim), **f32)
        out = L.RenderOut()
        out.rgb, out.depth, out.acc, out.feature = L.ptr(rgb), L.ptr(depth), L.ptr(acc), L.ptr(feat)
        out.feature_kind = 2 if self.extract_feature == "before_linear" else 1
        wts = None
        if self.mode == "nerf" and c.composite_feature:           # nerf.py:167-169: feature_map = sum_s w feature
            wts = torch.empty((R, S), **f32)
            out.weights = L.ptr(wts)
        ws, need = self._workspace(cfg, R)
        tr = t_rand.contiguous().float() if t_rand is not None else None
        n0 = noise0.contiguous().float() if noise0 is not None else None
        fn_rays, fn_render = ((L.lib().evd_nerf_render_rays, L.lib().evd_nerf_render) if self.mode == "nerf" else
                              (L.lib().evd_c2f_render_rays, L.lib().evd_c2f_render))
        if _rays is None:
            rb = ray_batch.contiguous().float()
            L.check(fn_rays(c.handle, None, C.byref(cfg), L.ptr(rb), R, L.ptr(tr), None, L.ptr(n0), None, C.byref(out), L.ptr(ws), need,
                            L.stream_ptr()), "coarse_render_rays")
        else:
            L.check(fn_render(c.handle, None, C.byref(cfg), L.ptr(_rays), R, L.ptr(tr), None, L.ptr(n0), None, C.byref(out), L.ptr(ws), need,
                              L.stream_ptr()), "coarse_render")
        if wts is not None:
            feat = (wts[..., None] * feat).sum(1)
        return rgb, feat

    def coarse_render(self, H, W, K, chunk=1 << 22, rays=None, c2w=None, ndc=True, near=0., far=1., use_viewdirs=False,
                      c2w_staticcam=None, **kwargs):
        """renderer.py:468-523: rays -> (rgb [R,3], feat) of the coarse level (what the PBE blur kernel consumes, renderer.py:296)."""
        if bool(use_viewdirs) != self.use_viewdirs:
            raise L.EvdError(f"coarse_render(use_viewdirs={use_viewdirs}) on a model built with use_viewdirs={self.use_viewdirs}")
        if c2w_staticcam is not None:
            raise NotImplementedError("coarse_render: c2w_staticcam goes through render()")
        if rays is None:
            if c2w is None:
                raise L.EvdError("coarse_render needs rays or c2w")
            o, d = get_rays(H, W, K, torch.as_tensor(c2w, device=self.device))
            rays = torch.stack([o, d], dim=-1)
        flat = rays.to(self.device).float().reshape(-1, 3, 2).contiguous()
        cfg = self._cfg(H, W, float(K[0][0]), ndc, near, far, kwargs.get("N_samples"), 0, kwargs.get("lindisp", False),
                        kwargs.get("perturb", 0.), kwargs.get("white_bkgd", False))
        rgbs, feats = [], []
        for i in range(0, max(1, flat.shape[0]), chunk):
            kw = {k: v for k, v in kwargs.items() if k in ("N_samples", "retraw", "lindisp", "perturb", "N_importance", "white_bkgd", "raw_noise_std",
                                                             "force_naive", "inference")}
            for k in ("t_rand", "noise0"):
                if kwargs.get(k) is not None:
                    kw[k] = kwargs[k][i:i + chunk]
            r, f = self.coarse_render_rays(None, _cfg=cfg, _rays=flat[i:i + chunk], **kw)
            rgbs.append(r)
            feats.append(f)
        return (rgbs[0], feats[0]) if len(rgbs) == 1 else (torch.cat(rgbs, 0), torch.cat(feats, 0))

    # ------------------------------------------------------------------ render_path, renderer.py:594-626
    def render_path(self, H, W, K, chunk, render_poses, render_kwargs, render_factor=0, shard_rows=False):
        """Full-frame renders of ``render_poses``.  ``shard_rows=True`` (new; the reference is single-GPU) splits the H image
        rows contiguously over the ranks of the initialised process group and all-gathers the row tiles (dist.gather_rows),
        so every rank returns the whole frames (BASELINE config 5)."""
        from . import dist as D
        if render_factor != 0:
            H, W = H // render_factor, W // render_factor
        lo, hi = 0, H
        if shard_rows:
            import torch.distributed as tdist
            if tdist.is_available() and tdist.is_initialized():
                lo, hi = D.shard_range(H, tdist.get_rank(), tdist.get_world_size())
        rgbs, depths = [], []
        for c2w in render_poses:
            dev = c2w.device if isinstance(c2w, torch.Tensor) else torch.device("cpu")
            o, d = get_rays(H, W, K, c2w if isinstance(c2w, torch.Tensor) else torch.as_tensor(c2w))
            rays = torch.stack([o, d], dim=-1)[lo:hi]
            rgb, depth, acc, extras = self.render(H, W, K, chunk=chunk, rays=rays, **render_kwargs)
            if shard_rows:
                rgb, depth = D.gather_rows(rgb, H), D.gather_rows(depth, H)
            rgbs.append(rgb.to(dev))
            depths.append(depth.to(dev))
        return torch.stack(rgbs, 0), torch.stack(depths, 0)

    # ------------------------------------------------------------------ forward, renderer.py:266-397
    # ---- the reference's nn.Module surface for a training loop (optimizer groups run_nerf.py:245-263, iteration :423-613,
    # ---- checkpoint :617-638) ---------------------------------------------------------------------------------------------
    def enable_training(self, state_dict, grads_in_place=False):
        """Create the trainable tensors from a state dict and keep them in the model: ONE LEAF PER REFERENCE PARAMETER
        (`named_parameters()` yields the reference's names), so that `parameters()`, `get_parameters(type, match_re, not_match_re)`,
        `grad_vars` and `grad_vars_vol` feed optimizer groups exactly like renderer.py:58-79,112-127 / run_nerf.py:245-263, and
        `model(H, W, K, chunk, rays=..., **render_kwargs_train)` in training mode runs forward_train (autograd).  The library's
        kernels consume one flat float32 tensor per network: the leaves are VIEWS of it (no concatenation per forward; the backward
        returns slices of the flat gradient); the tri-plane tensors are leaves in the library's channel-last layout ([H,W,C], [L,C];
        an element-wise optimizer does not care, `state_dict()` returns the reference layouts).

        grads_in_place=False (default): plain autograd semantics everywhere -- torch.autograd.grad, backward(inputs=...), hooks.
        grads_in_place=True: the backward kernels ADD the parameter and grid gradients straight into persistent buffers whose slices
        are the leaves' .grad, and return None to autograd.  Valid for the reference's loop only -- optimizer.zero_grad();
        loss.backward(); optimizer.step() (run_nerf.py:593-601) -- and what it saves per blurfactory iteration is ~90 gradient
        additions, ~50 copies and the zero-fill + re-add of 165 MB of grid gradients per scatter."""
        from collections import OrderedDict
        self._grads_in_place = bool(grads_in_place)
        levels = []
        for prefix, net, p in (("mlp_coarse.", self.mlp_coarse, None), ("mlp_fine.", self.mlp_fine, None)):
            if net is None:
                levels.append(None)
                continue
            flat = net.flat_params(state_dict, prefix, self.device).detach()
            blocks = sorted(net.param_blocks(), key=lambda b: b[2])
            off = 0
            for key, shape, o in blocks:             # the leaves tile the flat tensor exactly, in this order
                if o != off:
                    raise L.EvdError(f"parameter blocks of {prefix} are not contiguous at {key}")
                off += int(np.prod(shape))
            if off != flat.numel():
                raise L.EvdError(f"parameter blocks of {prefix} do not cover the flat tensor")
            grids = None
            if self.mode == "c2f":
                g = net.grid_params()
                grids = OrderedDict([(f"{prefix}app_plane.{i}", g[i]) for i in range(3)] + [(f"{prefix}app_line.{i}", g[3 + i]) for i in range(3)]
                                    + [(f"{prefix}basis_mat.weight", g[6])])
                net._grads_in_place = self._grads_in_place
                net._grid_grad_flat = (torch.empty((sum(t.numel() for t in g),), dtype=torch.float32, device=self.device)
                                       if self._grads_in_place else None)
            net._synced = net._synced_net = None
            absent = {key for key, _, _ in blocks if prefix + key not in state_dict}
            if any(not key.endswith(".bias") for key in absent):
                raise L.EvdError(f"state dict lacks {sorted(prefix + k for k in absent if not k.endswith('.bias'))}")
            levels.append(_Level(net, prefix, flat, blocks, grids, self._grads_in_place, absent))
        self._levels = levels
        return self

    def grad_buffers(self):
        """[(flat gradient buffer, [parameters whose .grad are its slices])] of a model in the in-place mode, for
        dist.GradReducer(flat_buffers=...): the collectives then run on the buffers themselves.  Empty in the default mode."""
        self._require_training()
        out = []
        for lv in self._levels:
            if lv is None or not lv.in_place:
                continue
            out.append((lv.attach_grads() if lv.flat_grad is None else lv.flat_grad, list(lv.leaves.values())))
            gbuf = getattr(lv.net, "_grid_grad_flat", None)
            if lv.grids is not None and gbuf is not None:
                out.append((gbuf, list(lv.grids.values())))
        return out

    def zero_grad(self, set_to_none=True):
        """drop (or zero) the gradients of every trainable tensor of the model, like nn.Module.zero_grad"""
        for p in self.parameters():
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()

    def _current_params(self):
        """(params_coarse, params_fine) in the form forward_train / render_rays_train take.  The flat tensors alias the levels' storage
        (same address, shared version counter): the library re-packs its weight streams / reloads its grids only after an optimizer
        step actually changed them, not once per forward."""
        return tuple(lv.params() if lv is not None else None for lv in self._levels)

    @property
    def _train_params(self):
        return self._current_params() if getattr(self, "_levels", None) is not None else None

    def _require_training(self):
        if getattr(self, "_levels", None) is None:
            raise L.EvdError("call enable_training(state_dict) first")

    def named_parameters(self):
        """[(reference state-dict name, leaf tensor)]: mlp_{coarse,fine}.* (+ kernelsnet.* / awpnet.* of the PyTorch modules)"""
        self._require_training()
        out = []
        for lv in self._levels:
            if lv is None:
                continue
            out += list(lv.leaves.items())
            if lv.grids is not None:
                out += list(lv.grids.items())
        for name, m in self._torch_children():
            out += [(f"{name}.{k}", v) for k, v in m.named_parameters()]
        return out

    def _torch_children(self):
        """the PyTorch modules NeRFAll holds in the reference (renderer.py:24-43), under the reference's attribute names; a FusedAWP
        wrapper is looked through, so parameter and state-dict keys stay `awpnet.sample_feature_embed_layer.0.weight` etc."""
        out = []
        for name, m in (("kernelsnet", self.kernelsnet), ("awpnet", self.awpnet)):
            m = getattr(m, "ref", m) if type(m).__name__ == "FusedAWP" else m
            if isinstance(m, torch.nn.Module):
                out.append((name, m))
        return out

    def parameters(self):
        return [v for _, v in self.named_parameters()]

    def get_parameters(self, type, match_re=None, not_match_re=None):
        """renderer.py:112-127: the 'net' or 'vol' (app_plane / app_line) parameters whose names match / do not match a regex,
        e.g. get_parameters("net", match_re=r"\\.color_net\\.[0-9]+\\.weight") for the weight-decay group of run_nerf.py:246-250"""
        import re
        is_vol = lambda k: "app_plane" in k or "app_line" in k
        ok = lambda k: ((match_re is None or len(re.findall(match_re, k)) > 0) and
                        (not_match_re is None or not len(re.findall(not_match_re, k)) > 0))
        return [v for k, v in self.named_parameters() if ok(k) and ((type == "net" and not is_vol(k)) or (type == "vol" and is_vol(k)))]

    @property
    def grad_vars_vol(self):
        """renderer.py:60,79 / voxnerf.py:121: lines then planes of the coarse level, then of the fine level"""
        self._require_training()
        out = []
        for lv in self._levels:
            if lv is not None and lv.grids is not None:
                g = list(lv.grids.values())
                out += g[3:6] + g[0:3]
        return out

    @property
    def grad_vars(self):
        """renderer.py:60-64,77-78 / voxnerf.py:122-123: basis_mat, color_net, sigma_net of the coarse level, the kernelsnet's and
        the awpnet's parameters, then the fine level's"""
        self._require_training()
        def level(lv):
            if lv is None or lv.grids is None:
                return []
            names = lv.leaves
            return ([list(lv.grids.values())[6]] + [v for k, v in names.items() if ".color_net." in k] +
                    [v for k, v in names.items() if ".sigma_net." in k])
        out = level(self._levels[0])
        for m in (self.kernelsnet, self.awpnet):
            if isinstance(m, torch.nn.Module):
                out += list(m.parameters())
        return out + level(self._levels[1])

    def state_dict(self):
        """current values under the reference's state-dict keys and layouts (checkpointing, run_nerf.py:617-638)"""
        self._require_training()
        sd = {}
        for name, net, lv in (("mlp_coarse.", self.mlp_coarse, self._levels[0]), ("mlp_fine.", self.mlp_fine, self._levels[1])):
            if lv is None:
                continue
            sd.update({k: v.detach().clone() for k, v in lv.leaves.items()})
            if lv.grids is not None:
                sd.update(net.grids_to_state_dict(list(lv.grids.values()), name))
        for name, m in self._torch_children():              # the reference's NeRFAll is an nn.Module: its state dict carries these too
            sd.update({f"{name}.{k}": v.detach().clone() for k, v in m.state_dict().items()})
        return sd

    def forward(self, H, W, K, chunk=1 << 22, rays=None, rays_info=None, poses=None, **kwargs):
        if not self.training:
            assert poses is not None, "Please specify poses when in the eval model"
            return self.render_path(H, W, K, chunk, poses, **kwargs)
        assert rays is not None, "Please specify rays when in the training mode"
        if getattr(self, "_levels", None) is not None:              # differentiable path (enable_training)
            kw = dict(kwargs)
            # keys of the reference's render_kwargs_train / call site (run_nerf.py:306-314,438-442) that select nothing here:
            # retraw (the dict always carries weights / z_vals), return_pts0_rgb (always returned), pytest, c2w, use_viewdirs
            # (networks are built with view directions); `inference` gates the AWP feature output as in renderer.py:255
            for k in ("retraw", "use_viewdirs", "c2w", "c2w_staticcam", "pytest", "return_pts0_rgb"):
                kw.pop(k, None)
            if kw.pop("inference", False):
                raise L.EvdError("inference=True in training mode: use eval() + render_path / render (renderer.py:394-397)")
            pc, pf = self._current_params()
            return self.forward_train(H, W, K, rays, pc, pf, rays_info=rays_info, force_naive=kw.pop("force_naive", True), **kw)
        force_baseline = kwargs.pop("force_naive", True)
        return_pts0_rgb = kwargs.pop("return_pts0_rgb", False)
        N_importance = kwargs.get("N_importance", 0)
        other_loss, other_tensors = {}, {}
        if self.kernelsnet is not None and not force_baseline:
            if self.kernel_type != "RBK":
                raise NotImplementedError("only the RBK kernel of the shipped configs is supported")
            from .losses import weighted_sum
            new_rays, weight1, align_loss, extra1 = self.kernelsnet(H, W, K, rays, rays_info, feats=None,
                                                                    return_img_embed=self.use_awp)
            extra1 = {f"stage1_{k}": v for k, v in extra1.items()}
            ray_num, pt_num = new_rays.shape[:2]
            rgb, depth, acc, extras = self.render(H, W, K, chunk, new_rays.reshape(-1, 3, 2), **kwargs)
            rgb_pts = rgb.reshape(ray_num, pt_num, 3)
            rgb1_pts = extras["rgb0"].reshape(ray_num, pt_num, 3) if N_importance > 0 else None
            if self.use_awp:
                ccw_fine = self.awpnet(extras["depth_feature"], extras["z_vals"], extras["rays_d"], extra1["stage1_img_embed"])
                ccw_fine = ccw_fine + ccw_fine * self.awpnet.ccw_fine_scale
                ccw_fine = ccw_fine / torch.sum(ccw_fine, -1, keepdim=True)
                other_tensors["rgb_awp"] = weighted_sum(rgb, ccw_fine)
            rgb_out = weighted_sum(rgb, weight1)
            rgb1 = weighted_sum(extras["rgb0"], weight1) if N_importance > 0 else None
            self._tv(other_loss, N_importance)
            if align_loss is not None:
                other_loss["align"] = align_loss.reshape(1, 1)
            other_tensors.update(extra1)
            if return_pts0_rgb:
                other_tensors["stage1_rgb_pts0"] = rgb_pts[:, 0]
                if N_importance > 0:
                    other_tensors["stage1_rgb1_pts0"] = rgb1_pts[:, 0]
            return rgb_out, rgb1, other_loss, other_tensors
        rgb, depth, acc, extras = self.render(H, W, K, chunk, rays, **kwargs)
        other_tensors["stage1_rgb_pts0"] = rgb
        if N_importance > 0:
            other_tensors["stage1_rgb1_pts0"] = extras["rgb0"]
        self._tv(other_loss, N_importance)
        return rgb, extras["rgb0"] if "rgb0" in extras else None, other_loss, other_tensors

    def _tv(self, other_loss, N_importance):
        """TV regulariser of the tri-planes, renderer.py:361-365 / :385-389."""
        if self.mode == "c2f":
            tv = self.mlp_coarse.TV_loss_app()
            if N_importance > 0 and self.mlp_fine is not None:
                tv = tv + self.mlp_fine.TV_loss_app()
            other_loss["TV"] = tv * 5

    __call__ = forward
