"""Mirror of the reference ``networks/pdrf/voxnerf.py`` (VoxelNeRFBase :6, VoxelNeRFRayFeatures :262,
VoxelNeRFSampleFeatures :284) on libevdnerf.so: tri-plane feature gather, sigma/colour MLPs, TV regulariser."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _lib as L
from .weights import pdrf_grid_size


def _np32(v):
    if isinstance(v, torch.Tensor):
        v = v.detach().cpu().numpy()
    return np.ascontiguousarray(v, dtype=np.float32)


# Grid gradients of the training path.  DEFAULT: plain autograd returns (torch.autograd.grad, backward(inputs=...), tensor hooks and
# post-accumulate-grad hooks all work on the grid leaves).  OPT-IN (set_grads_in_place(True), or NeRFAll.enable_training(...,
# grads_in_place=True), or EVD_GRADS_IN_PLACE=1): the scatter / TV backward kernels add straight into the leaves' .grad and return
# None to autograd -- valid ONLY for a plain loss.backward() followed by optimizer.step() (what run_nerf.py:593-601 does); it saves
# zeroing and re-adding 165 MB of gradient tensors eleven times per blurfactory iteration.
_GRADS_IN_PLACE = os.environ.get("EVD_GRADS_IN_PLACE", "0") == "1"


def set_grads_in_place(on: bool):
    """Opt in to / out of the in-place accumulation of the grid gradients (see above).  Returns the previous setting."""
    global _GRADS_IN_PLACE
    prev, _GRADS_IN_PLACE = _GRADS_IN_PLACE, bool(on)
    return prev


def grads_in_place() -> bool:
    return _GRADS_IN_PLACE


def _grid_grads(net, like, in_place=False):
    """gradient tensors for the 7 grid parameters + the ctypes struct pointing at them.  The scatter / TV backward kernels ADD into
    what they are given.  in_place (every grid a leaf tensor, float32, contiguous): they add straight into the leaves' .grad -- a
    whole blurfactory iteration runs 9 scatters and 2 TV backwards, each of which would otherwise zero 7 fresh tensors (165 MB for
    the fine level) and have autograd add them to .grad again; the returned list is then all None (nothing left for autograd to do)."""
    if in_place and all(t.is_leaf and t.requires_grad and t.dtype == torch.float32 and t.is_contiguous() for t in like):
        buf = getattr(net, "_grid_grad_flat", None)
        if buf is not None and all(t.grad is None for t in like) and buf.numel() == sum(t.numel() for t in like):
            buf.zero_()                 # ONE fill for the level's seven gradient tensors (165 MB for the fine level), then views of it
            off = 0
            for t in like:
                t.grad = buf[off:off + t.numel()].view(t.shape)
                off += t.numel()
        for t in like:
            if t.grad is None:
                t.grad = torch.zeros_like(t)
        tgt, ret = [t.grad for t in like], [None] * len(like)
        if not all(g.is_contiguous() and g.dtype == torch.float32 for g in tgt):
            return _grid_grads(net, like, False)
    else:
        tgt = ret = [torch.zeros_like(t) for t in like]
    gs = L.VoxelGridGrads()
    for i in range(3):
        gs.plane[i], gs.line[i] = tgt[i].data_ptr(), tgt[3 + i].data_ptr()
    gs.basis = tgt[6].data_ptr()
    return ret, gs


# ---- "this level's gradients are complete" (data-parallel training, dist.GradReducer.attach): every training-forward call of a level
# (gather, networks, TV) counts one pending backward node; the node that brings the count back to zero fires the level's callback --
# its gradient buffers can be all-reduced while the backward of the levels / modules in front of it still runs.
def _fwd_noted(net):
    if torch.is_grad_enabled() and getattr(net, "_grads_ready_cb", None) is not None:
        guard = getattr(net, "_fwd_guard_cb", None)
        if guard is not None:
            guard()                 # dist.GradReducer: raises if this level's all-reduce of the running step is already in flight
        net._pending_bwd = getattr(net, "_pending_bwd", 0) + 1


def _bwd_done(net):
    cb = getattr(net, "_grads_ready_cb", None)
    if cb is None:
        return
    net._pending_bwd = getattr(net, "_pending_bwd", 0) - 1
    if net._pending_bwd == 0:
        cb()


class _VoxelSample(torch.autograd.Function):
    """VoxelNeRFBase.sample with gradients to the (channel-last) planes, lines and basis_mat: evd_voxel_sample / _bwd"""

    @staticmethod
    def forward(ctx, pts, net, precision, *grids):
        ctx.net, ctx.pts, ctx.precision = net, pts, precision
        ctx.save_for_backward(*grids)
        out, net._sample_out = getattr(net, "_sample_out", None), None        # a strided window to write into (sample_train(out=...))
        return net.sample(pts, precision, out=out)

    @staticmethod
    def backward(ctx, d_out):
        L.note_backward()
        net, pts = ctx.net, ctx.pts.reshape(-1, 3).contiguous().float()
        # the gradient rows as they arrive: a window of a wider row buffer (renderer._MergeFeatures) is read in place through its row stride
        if d_out.dtype == torch.float32 and d_out.dim() == 3 and d_out.stride(2) == 1 and d_out.stride(0) == d_out.shape[1] * d_out.stride(1):
            g, g_stride = d_out, d_out.stride(1)
        else:
            g, g_stride = d_out.reshape(-1, net.app_dim).contiguous().float(), net.app_dim
        grads, gs = _grid_grads(net, ctx.saved_tensors, in_place=getattr(net, "_grads_in_place", _GRADS_IN_PLACE) and not torch.is_grad_enabled())
        d_pts = torch.empty_like(pts) if ctx.needs_input_grad[0] else None
        # scratch for the hybrid form of the scatter (csrc/kernel_voxel_scatter.hip: plane taps by direct float atomics, line taps through
        # fixed-point LDS slices -- a third fewer atomic requests, 24-27 % faster); EVD_SCATTER=direct passes none: every tap an atomic
        nb = int(L.lib().evd_voxel_sample_bwd_workspace_bytes(net._h, pts.shape[0])) if os.environ.get("EVD_SCATTER") != "direct" else 0
        ws = torch.empty((nb,), dtype=torch.uint8, device=pts.device) if nb else None
        # in the forward's arithmetic mode: where it interpolated the float16 grid copies, the re-gather of the backward reads them too
        prec = L.PREC[ctx.precision] if ctx.precision is not None else L.PREC["f32"]
        L.check(L.lib().evd_voxel_sample_bwd_prec(net._h, prec, L.ptr(pts), pts.shape[0], C.c_void_p(g.data_ptr()), g_stride, 0, C.byref(gs), L.ptr(d_pts), L.ptr(ws), nb,
                                                  L.stream_ptr()), "evd_voxel_sample_bwd_prec")
        _bwd_done(net)
        return (d_pts.reshape(ctx.pts.shape) if d_pts is not None else None, None, None, *grads)


class GeoFragments:
    """The fine level's per-sample geo features of ONE training forward, left where the level's kernel keeps them for its own
    backward: float16 / bfloat16 MFMA fragments in its activation store.  This is what `depth_feature` (renderer.py:253-256) becomes
    when the AWP consumer is the fused one (awp.FusedAWP): the float32 tensor [R P, S, 128] is never written.  `token` ties the
    consumer into the autograd graph; `awp_store` is set by the consumer's backward (its d geo fragments, picked up by
    evd_voxel_mlp_backward)."""

    def __init__(self):
        self.level = self.store = self.token = self.awp_store = self.precision = None
        self.R = self.S = 0


class _VoxelMLP(torch.autograd.Function):
    """per-sample sigma / colour networks of one level (voxnerf.py:210-221,240-254) with the hand-written backward:
    gradients to the flat parameter tensor and to the sampled features (evd_voxel_mlp_train / _backward)"""

    @staticmethod
    def forward(ctx, flat, fts, pts, viewdirs, net, precision, want_feature=False):
        geo = want_feature if isinstance(want_feature, GeoFragments) else None
        rows = bool(want_feature) and geo is None
        raw, store, feature = net.mlpforward_train(pts, viewdirs, fts, precision, want_feature=rows)
        ctx.net, ctx.precision, ctx.store, ctx.raw = net, precision, store, raw
        ctx.ft_shape, ctx.pts, ctx.viewdirs, ctx.has_feature, ctx.geo = fts.shape, pts, viewdirs, rows, geo
        ctx.accum = getattr(flat, "_evd_accum", None)          # in-place gradient accumulation (renderer._FlatParams), opt-in
        ctx.set_materialize_grads(False)
        if geo is not None:
            # (the mixed training modes f16c / f16m keep the float16 mode's store: that is the fragment format a consumer reads)
            geo.level, geo.store, geo.R, geo.S = net, store, pts.shape[0], pts.shape[1]
            geo.precision = "f16" if precision in ("f16c", "f16m") else precision
            return raw, torch.zeros((1,), dtype=torch.float32, device=raw.device)
        return (raw, feature) if rows else raw

    @staticmethod
    def backward(ctx, d_raw, d_feature=None):
        L.note_backward()
        need = ctx.needs_input_grad
        if d_raw is None:
            d_raw = torch.zeros_like(ctx.raw)
        awp_store = ctx.geo.awp_store if ctx.geo is not None else None
        acc = ctx.accum() if (ctx.accum is not None and need[0] and not torch.is_grad_enabled()) else None
        gflat, d_fts, d_pts, d_dirs = ctx.net.mlp_backward_flat(d_raw, ctx.raw, ctx.store, ctx.precision, want_fts=need[1],
                                                                pts=ctx.pts if need[2] else None, viewdirs=ctx.viewdirs if need[3] else None,
                                                                d_feature=d_feature if ctx.has_feature else None, awp_store=awp_store,
                                                                accumulate_into=acc)
        ctx.store = ctx.raw = None
        if ctx.geo is not None:
            ctx.geo.store = ctx.geo.awp_store = None
        _bwd_done(ctx.net)
        R, S = ctx.pts.shape[:2]
        return (gflat, d_fts.reshape(ctx.ft_shape) if d_fts is not None else None, d_pts.reshape(ctx.pts.shape) if d_pts is not None else None,
                d_dirs.reshape(R, S, 3).sum(1) if d_dirs is not None else None, None, None, None)


class _VoxelTV(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, *grids):
        ctx.net = net
        ctx.save_for_backward(*grids)
        return net.TV_loss_app()

    @staticmethod
    def backward(ctx, d_loss):
        L.note_backward()
        net = ctx.net
        grads, gs = _grid_grads(net, ctx.saved_tensors, in_place=getattr(net, "_grads_in_place", _GRADS_IN_PLACE) and not torch.is_grad_enabled())
        gs.basis = None
        d = d_loss.reshape(1).contiguous().float()
        L.check(L.lib().evd_voxel_tv_loss_bwd(net._h, L.ptr(d), C.byref(gs), L.stream_ptr()), "evd_voxel_tv_loss_bwd")
        _bwd_done(net)
        return (None, *grads)


class VoxelNeRFBase:
    def __init__(self, state_dict, prefix, aabb, num_layers=2, hidden_dim=64, geo_feat_dim=15, num_layers_color=3,
                 hidden_dim_color=64, input_ch=95, multires=10, multires_views=4, render_rmnearplane=0, app_dim=32,
                 app_n_comp=(64, 16, 16), n_voxels=134217984, rgb_activate="sigmoid", sigma_activate="relu",
                 composite_feature=False, app_actfn="none", precision="f16x3"):
        lo, hi = [float(v) for v in aabb[0]], [float(v) for v in aabb[1]]
        self.gridSize = pdrf_grid_size(lo, hi, n_voxels)
        self.hidden_dim, self.geo_feat_dim, self.app_dim = hidden_dim, geo_feat_dim, app_dim
        self.input_ch, self.precision = input_ch, precision
        self.ft_dim = input_ch - 3 * (1 + 2 * multires)
        self.rgb_activate, self.sigma_activate = rgb_activate, sigma_activate
        self.render_rmnearplane = render_rmnearplane
        self.composite_feature = composite_feature
        self.training = False
        g = lambda k: _np32(state_dict[prefix + k]) if (prefix + k) in state_dict else None
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None
        self.has_color_bias = (prefix + "color_net.0.bias") in state_dict          # voxnerf.py:80 add_bias_color (off in every shipped config)
        keep = []
        d = L.VoxelDesc()
        d.num_layers, d.hidden_dim, d.geo_feat_dim, d.num_layers_color = num_layers, hidden_dim, geo_feat_dim, num_layers_color
        d.input_ch, d.multires, d.multires_views, d.app_dim = input_ch, multires, multires_views, app_dim
        for i in range(3):
            d.n_comp[i], d.grid[i] = int(app_n_comp[i]), int(self.gridSize[i])
            d.aabb[i], d.aabb[3 + i] = lo[i], hi[i]
        d.app_act, d.rgb_act, d.sigma_act = L.ACT[app_actfn], L.ACT[rgb_activate], L.ACT[sigma_activate]
        d.composite_feature, d.rmnear = int(bool(composite_feature)), float(render_rmnearplane)
        for l in range(num_layers):
            w = g(f"sigma_net.{l}.weight")
            keep.append(w)
            d.sigma_w[l] = fp(w)
        for l in range(num_layers_color):
            w, b = g(f"color_net.{l}.weight"), g(f"color_net.{l}.bias")
            keep += [w, b]
            d.color_w[l], d.color_b[l] = fp(w), fp(b)
        mat, vec = [[0, 1], [0, 2], [1, 2]], [2, 1, 0]
        for i in range(3):
            pl, li = g(f"app_plane.{i}"), g(f"app_line.{i}")
            exp_p = (1, app_n_comp[i], self.gridSize[mat[i][1]], self.gridSize[mat[i][0]])
            if pl is None or tuple(pl.shape) != exp_p:
                raise L.EvdError(f"{prefix}app_plane.{i}: expected shape {exp_p}, got {None if pl is None else pl.shape}")
            keep += [pl, li]
            d.plane[i], d.line[i] = fp(pl), fp(li)
        b = g("basis_mat.weight")
        keep.append(b)
        d.basis = fp(b)
        h = C.c_void_p()
        L.check(L.lib().evd_voxel_create(C.byref(d), C.byref(h)), "evd_voxel_create")
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and L is not None and getattr(L, "lib", None) is not None:     # module globals may be gone at interpreter shutdown
            L.lib().evd_voxel_destroy(h)
            self._h = None

    @property
    def handle(self):
        return self._h

    def train(self, mode=True):
        self.training = mode
        return self

    # voxnerf.py:153-201; returns the reference 5-tuple (rgb_map, density, acc_map, weights, depth_map).  Sigma is channel 0, the
    # (already sigmoided) colour channels 1..3 (:172,179); rgb_activate is relu (coarse level) or none (fine level).
    def raw2outputs(self, raw, z_vals, rays_d, raw_noise_std=0., is_train=False, noise=None):
        from .nerf import _Raw2Outputs
        raw, z, rd = raw.contiguous().float(), z_vals.contiguous().float(), rays_d.contiguous().float()
        R, S, Cc = raw.shape
        if raw_noise_std > 0. and noise is None:
            noise = torch.randn((R, S - 1), dtype=torch.float32, device=raw.device) * raw_noise_std
        nz = noise.contiguous().float() if noise is not None else None
        thr = float(self.render_rmnearplane) / 128.0 if (not is_train and self.render_rmnearplane > 0) else 0.0
        if Cc != 4:           # composite_feature=True (PBE): raw = [sigma | geo features], the map has Cc - 1 channels (:223-229); forward only
            f32 = dict(dtype=torch.float32, device=raw.device)
            fmap, dens, acc = torch.empty((R, Cc - 1), **f32), torch.empty((R, S - 1), **f32), torch.empty((R,), **f32)
            wts, depth = torch.empty((R, S), **f32), torch.empty((R,), **f32)
            L.check(L.lib().evd_raw2outputs(L.ptr(raw.detach()), L.ptr(z), L.ptr(rd), rd.shape[-1], R, S, Cc, 0, 1, Cc - 1, L.ACT[self.rgb_activate],
                                            L.ACT[self.sigma_activate], 0, thr, L.ptr(nz), L.ptr(fmap), L.ptr(dens), L.ptr(acc), L.ptr(wts), L.ptr(depth),
                                            None, 0, None, L.stream_ptr()), "evd_raw2outputs")
            return fmap, dens, acc, wts, depth
        rgb, dens, acc, wts, depth = _Raw2Outputs.apply(raw, z, rd, nz, 0, 1, L.ACT[self.rgb_activate], L.ACT[self.sigma_activate], False, thr)
        return rgb, dens, acc, wts, depth

    # voxnerf.py:203-208
    def sample(self, pts, precision=None, out=None):
        """voxnerf.py:203-208.  precision None: the float32 grids (the reference's arithmetic); a mode name: as the c2f renderer samples
        in that mode (f16 / bf16 read the float16 copies of the grids).  out: a float32 [R, S, app_dim] tensor to write instead of a new
        one -- rows of unit channel stride and ONE row stride (a column window of a wider row buffer)"""
        sh = pts.shape
        p = pts.reshape(-1, 3).contiguous().float()
        if out is not None:
            if not (len(sh) == 3 and out.dtype == torch.float32 and out.shape == (sh[0], sh[1], self.app_dim) and out.stride(2) == 1
                    and out.stride(0) == sh[1] * out.stride(1) and out.stride(1) >= self.app_dim):
                raise L.EvdError("sample(out=...): [R, S, app_dim] float32 rows with unit channel stride and one row stride")
            dst, stride = out, out.stride(1)
        else:
            dst, stride = torch.empty((p.shape[0], self.app_dim), dtype=torch.float32, device=p.device), self.app_dim
        if precision is None:
            L.check(L.lib().evd_voxel_sample(self._h, L.ptr(p), p.shape[0], C.c_void_p(dst.data_ptr()), stride, 0, L.stream_ptr()), "evd_voxel_sample")
        else:
            L.check(L.lib().evd_voxel_sample_prec(self._h, L.PREC[precision], L.ptr(p), p.shape[0], C.c_void_p(dst.data_ptr()), stride, 0, L.stream_ptr()),
                    "evd_voxel_sample_prec")
        if out is not None:
            return out
        return dst.reshape(sh[0], sh[1], self.app_dim) if len(sh) == 3 else dst

    # ---- training the sigma / colour networks (f16 / bf16): one flat float32 parameter tensor, library order ----------------
    PARAM_KEYS = ["sigma_net.0.weight", "sigma_net.1.weight", "color_net.0.weight", "color_net.0.bias", "color_net.1.weight",
                  "color_net.1.bias", "color_net.2.weight", "color_net.2.bias"]

    def param_blocks(self):
        if getattr(self, "_blocks", None) is None:
            off = (C.c_long * 9)()
            if L.lib().evd_voxel_param_blocks(self._h, off, 9) != 8:
                raise L.EvdError("evd_voxel_param_blocks: unexpected parameter block count")
            HD, G = self.hidden_dim, self.geo_feat_dim
            outs = [HD, 1 + G, HD, HD, HD, HD, 3, 3]
            blocks = []
            for i, key in enumerate(self.PARAM_KEYS):
                n = off[i + 1] - off[i]
                blocks.append((key, (outs[i], n // outs[i]) if key.endswith("weight") else (n,), off[i]))
            self._blocks, self._nparam = blocks, off[8]
        return self._blocks

    def flat_params(self, state_dict, prefix="", device="cuda"):
        blocks = self.param_blocks()
        flat = torch.zeros((self._nparam,), dtype=torch.float32)
        for key, shape, off in blocks:
            if prefix + key in state_dict:
                flat[off:off + int(np.prod(shape))] = torch.as_tensor(_np32(state_dict[prefix + key])).reshape(-1)
        return flat.to(device).requires_grad_(True)

    def unflatten(self, flat):
        return {key: flat[off:off + int(np.prod(shape))].view(shape) for key, shape, off in self.param_blocks()}

    def load_params(self, flat):
        f = flat.detach().contiguous().float()
        self.param_blocks()
        if f.numel() != self._nparam:
            raise L.EvdError(f"flat parameter tensor has {f.numel()} elements, the level {self._nparam}")
        L.check(L.lib().evd_voxel_load_params(self._h, L.ptr(f), L.stream_ptr()), "evd_voxel_load_params")
        self._synced_net = (flat.data_ptr(), flat._version, L.backward_generation())

    def mlpforward_train(self, pts, viewdirs, fts, precision=None, want_feature=False):
        p, vd, ft = pts.contiguous().float(), viewdirs.contiguous().float(), fts.contiguous().float()
        R, S = p.shape[:2]
        raw = torch.empty((R, S, 4), dtype=torch.float32, device=p.device)
        feature = torch.empty((R, S, self.geo_feat_dim), dtype=torch.float32, device=p.device) if want_feature else None
        nb = int(L.lib().evd_voxel_train_store_bytes_prec(self._h, L.PREC[precision or self.precision], R * S))
        if nb == 0 and R * S > 0:
            raise L.EvdError(f"the training path is built for precision f16 / bf16 / f16x3 / f16c / f16m, not {precision or self.precision}")
        store = torch.empty((nb,), dtype=torch.uint8, device=p.device)
        L.check(L.lib().evd_voxel_mlp_train(self._h, L.PREC[precision or self.precision], L.ptr(p), L.ptr(vd), vd.shape[-1], L.ptr(ft), ft.shape[-1],
                                             R, S, L.ptr(raw), L.ptr(feature), L.ptr(store), nb, L.stream_ptr()), "evd_voxel_mlp_train")
        return raw, store, feature

    def mlp_backward_flat(self, d_raw, raw, store, precision=None, want_fts=True, pts=None, viewdirs=None, d_feature=None, awp_store=None,
                          accumulate_into=None):
        """-> (flat parameter gradient, d fts | None, d pts | None, d dirs per sample | None); the last two (through the positional
        encodings) are computed when the forward's pts / viewdirs are passed.  accumulate_into: a persistent flat float32 gradient buffer
        the kernels ADD into (evd_voxel_grads.accumulate); the returned flat gradient is then None"""
        g = d_raw.contiguous().float()
        R, S = g.shape[:2]
        blocks = self.param_blocks()
        flat = accumulate_into if accumulate_into is not None else torch.zeros((self._nparam,), dtype=torch.float32, device=g.device)
        gs, base = L.VoxelGrads(), flat.data_ptr()
        gs.accumulate = int(accumulate_into is not None)
        for key, shape, off in blocks:
            name, idx, kind = key.split(".")
            if kind == "bias" and not self.has_color_bias:
                continue            # rgb_add_bias off: the reference model has no such parameter -- a NULL gradient pointer lets the library skip the bias column
            arr = getattr(gs, ("sigma_" if name == "sigma_net" else "color_") + ("w" if kind == "weight" else "b"))
            arr[int(idx)] = base + 4 * off
        d_fts = torch.empty((R * S, self.ft_dim), dtype=torch.float32, device=g.device) if want_fts else None      # every column is written (k_frags_to_rows)
        nb = int(L.lib().evd_voxel_backward_workspace_bytes())
        ws = torch.empty((nb,), dtype=torch.uint8, device=g.device)
        p = pts.contiguous().float() if pts is not None else None
        vd = viewdirs.contiguous().float() if viewdirs is not None else None
        d_pts = torch.empty((R * S, 3), dtype=torch.float32, device=g.device) if p is not None else None
        d_dirs = torch.empty((R * S, 3), dtype=torch.float32, device=g.device) if vd is not None else None
        df = d_feature.contiguous().float() if d_feature is not None else None
        L.check(L.lib().evd_voxel_mlp_backward(self._h, L.PREC[precision or self.precision], L.ptr(g), L.ptr(raw), L.ptr(df), L.ptr(awp_store),
                                                awp_store.numel() if awp_store is not None else 0, R, S, L.ptr(store), store.numel(),
                                                C.byref(gs), L.ptr(d_fts), self.ft_dim, L.ptr(p), L.ptr(vd), vd.shape[-1] if vd is not None else 0,
                                                L.ptr(d_pts), L.ptr(d_dirs), L.ptr(ws), nb, L.stream_ptr()), "evd_voxel_mlp_backward")
        return (None if accumulate_into is not None else flat), d_fts, d_pts, d_dirs

    def mlp_train(self, flat, pts, viewdirs, fts, precision=None, want_feature=False):
        """raw [R,S,4] = (sigma, sigmoid(colour)) with autograd to the flat parameters, the sampled features and the rays;
        want_feature (fine level): also the per-sample geo features [R,S,geo] (voxnerf.py:221), an autograd output too; a
        GeoFragments instance instead of True: the features stay fragments in the level's store, the second output is its token"""
        if getattr(self, "_synced_net", None) != (flat.data_ptr(), flat._version, L.backward_generation()):
            self.load_params(flat)
        _fwd_noted(self)
        return _VoxelMLP.apply(flat, fts, pts, viewdirs, self, precision or self.precision, want_feature)

    # ---- training the grids (the library's channel-last layout; a permute away from the state dict) ---------------------
    def grid_params(self):
        """[plane0..2 [H,W,C], line0..2 [L,C], basis [app_dim, sum C]] as leaf tensors for an optimizer"""
        sz = (C.c_long * 7)()
        L.check(L.lib().evd_voxel_grid_sizes(self._h, sz), "evd_voxel_grid_sizes")
        mat, vec, g = [[0, 1], [0, 2], [1, 2]], [2, 1, 0], self.gridSize
        shapes = [(g[mat[i][1]], g[mat[i][0]], sz[i] // (g[mat[i][1]] * g[mat[i][0]])) for i in range(3)]
        shapes += [(g[vec[i]], sz[3 + i] // g[vec[i]]) for i in range(3)]
        shapes += [(self.app_dim, sz[6] // self.app_dim)]
        ts = [torch.empty(sh, dtype=torch.float32, device="cuda") for sh in shapes]
        pl, li = (C.c_void_p * 3)(*[t.data_ptr() for t in ts[:3]]), (C.c_void_p * 3)(*[t.data_ptr() for t in ts[3:6]])
        L.check(L.lib().evd_voxel_get_grids(self._h, pl, li, L.ptr(ts[6]), L.stream_ptr()), "evd_voxel_get_grids")
        return [t.requires_grad_(True) for t in ts]

    def load_grids(self, grids):
        ts = [t.detach().contiguous().float() for t in grids]
        pl, li = (C.c_void_p * 3)(*[t.data_ptr() for t in ts[:3]]), (C.c_void_p * 3)(*[t.data_ptr() for t in ts[3:6]])
        L.check(L.lib().evd_voxel_load_grids(self._h, pl, li, L.ptr(ts[6]), L.stream_ptr()), "evd_voxel_load_grids")
        self._synced = tuple((t.data_ptr(), t._version) for t in grids) + (L.backward_generation(),)

    def _sync(self, grids):
        if getattr(self, "_synced", None) != tuple((t.data_ptr(), t._version) for t in grids) + (L.backward_generation(),):
            self.load_grids(grids)

    def sample_train(self, pts, grids, precision=None, out=None):
        """sample(pts) with autograd to the grid parameters (re-loads them into the library after an optimizer step).  precision: the
        arithmetic mode of the training forward -- f16 / bf16 gather the float16 grid copies exactly as the inference render of that mode
        does (the copies are refreshed by the re-load); None / the float32-grade modes read the float32 grids.  The backward (scatter-add,
        the interpolation-weight derivative) always works on the float32 grids."""
        self._sync(grids)
        _fwd_noted(self)
        self._sample_out = out              # taken (and cleared) by _VoxelSample.forward; not an autograd input: it is written through its pointer
        try:
            return _VoxelSample.apply(pts, self, precision, *grids)
        finally:
            self._sample_out = None

    def tv_loss_train(self, grids):
        self._sync(grids)
        _fwd_noted(self)
        return _VoxelTV.apply(self, *grids)

    @staticmethod
    def grids_to_state_dict(grids, prefix=""):
        """channel-last parameters -> reference layouts (app_plane.i [1,C,H,W], app_line.i [1,C,L,1], basis_mat.weight)"""
        sd = {}
        for i in range(3):
            sd[f"{prefix}app_plane.{i}"] = grids[i].detach().permute(2, 0, 1).unsqueeze(0).contiguous()
            sd[f"{prefix}app_line.{i}"] = grids[3 + i].detach().t().unsqueeze(0).unsqueeze(-1).contiguous()
        sd[f"{prefix}basis_mat.weight"] = grids[6].detach().clone()
        return sd

    # voxnerf.py:210-259; returns (color, depth_map, acc_map, weights, feature_map)
    def forward(self, pts, viewdirs, fts, z_vals, rays_d, raw_noise_std=0., is_train=False, precision=None):
        if raw_noise_std > 0:
            raise NotImplementedError("density noise goes through render_rays(noise0=..., noise1=...)")
        p = pts.contiguous().float()
        R, S = p.shape[:2]
        vd, ft, z, rd = viewdirs.contiguous().float(), fts.contiguous().float(), z_vals.contiguous().float(), rays_d.contiguous().float()
        dev = p.device
        f32 = dict(dtype=torch.float32, device=dev)
        color, depth, acc = torch.empty((R, 3), **f32), torch.empty((R,), **f32), torch.empty((R,), **f32)
        # (the compensated float16 mode is an inference mode without per-sample feature rows: feature_map is None there)
        # composite_feature (PBE): the returned feature map is the composited one [R, geo] (voxnerf.py:226)
        if self.composite_feature:
            wts, feat = torch.empty((R, S), **f32), torch.empty((R, self.geo_feat_dim), **f32)
        else:
            wts, feat = torch.empty((R, S), **f32), (torch.empty((R, S, self.geo_feat_dim), **f32) if (precision or self.precision) != "f16c" else None)
        need = int(L.lib().evd_voxel_forward_workspace_bytes(self._h, R, S))
        ws = torch.empty((need,), dtype=torch.uint8, device=dev)
        L.check(L.lib().evd_voxel_forward(self._h, L.PREC[precision or self.precision], L.ptr(p), L.ptr(vd), 3, L.ptr(ft), ft.shape[-1],
                                          L.ptr(z), L.ptr(rd), 3, R, S, int(bool(is_train)), L.ptr(color), L.ptr(depth), L.ptr(acc),
                                          L.ptr(wts), L.ptr(feat), L.ptr(ws), need, L.stream_ptr()), "evd_voxel_forward")
        return color, depth, acc, wts, feat

    __call__ = forward

    # voxnerf.py:126-130
    def TV_loss_app(self):
        out = torch.empty((1,), dtype=torch.float32, device="cuda")
        L.check(L.lib().evd_voxel_tv_loss(self._h, L.ptr(out), L.stream_ptr()), "evd_voxel_tv_loss")
        return out[0]


class VoxelNeRFRayFeatures(VoxelNeRFBase):
    def __init__(self, *a, n_voxels=16777248, rgb_activate="relu", **kw):
        super().__init__(*a, n_voxels=n_voxels, rgb_activate=rgb_activate, **kw)


class VoxelNeRFSampleFeatures(VoxelNeRFBase):
    def __init__(self, *a, n_voxels=134217984, rgb_activate="none", **kw):
        super().__init__(*a, n_voxels=n_voxels, rgb_activate=rgb_activate, **kw)
