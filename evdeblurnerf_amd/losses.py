"""Loss-side pixel ops of the path: rbk_weighted_sum (networks/dpnerf/blurmodel.py:112-127), img2mse
(utils/metrics.py:7), egm_loss (utils/events.py:260-284) and the two fused per-step reductions whose
packed partial sums are what data-parallel ranks all-reduce (spec: run_nerf.py:443-497, 518-591)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib as L

BLUR_PARTIALS = 8    # [se_rgb, se_rgb1, se_awp, se_pts0_fine, se_pts0_coarse, n_elem, -, -]
EVENT_PARTIALS = 4   # [sum w (pred-bii)^2 fine, same coarse, sum w, -]


def weighted_sum(x, ccw):
    """out[r] = sum_p ccw[r,p] x[r*P+p]; x [R*P, ...] -> [R, ...]."""
    ccw = ccw.contiguous().float()
    R, P = ccw.shape
    xx = x.contiguous().float()
    Cc = int(np.prod(xx.shape[1:])) if xx.ndim > 1 else 1
    out = torch.empty((R, Cc), dtype=torch.float32, device=xx.device)
    L.check(L.lib().evd_weighted_sum(L.ptr(xx), L.ptr(ccw), R, P, Cc, L.ptr(out), L.stream_ptr()), "evd_weighted_sum")
    return out.reshape((R,) + tuple(xx.shape[1:]))


def rbk_weighted_sum(rgb, depth, acc, extras, ccw):
    """networks/dpnerf/blurmodel.py:112-127 (same return tuple; every extras entry reduced)."""
    out = {k: weighted_sum(v, ccw) for k, v in extras.items()}
    return weighted_sum(rgb, ccw), weighted_sum(depth, ccw), weighted_sum(acc, ccw), out


def blur_loss_partials(crf_rgb, rgb_p, w1, target, rgb0_p=None, w2=None, target_pts0=None, skip_learn_crf=False,
                       partial=None, want_colours=False):
    """One launch for the whole image-loss block. Returns (partial [8], colours dict)."""
    w1 = w1.contiguous().float()
    R, P = w1.shape
    dev = w1.device
    if partial is None:
        partial = torch.zeros((BLUR_PARTIALS,), dtype=torch.float32, device=dev)
    c = lambda t: t.contiguous().float() if t is not None else None
    rgb_p, rgb0_p, w2, target, target_pts0 = c(rgb_p), c(rgb0_p), c(w2), c(target), c(target_pts0)
    cols = {}
    if want_colours:
        cols = {"rgb": torch.empty((R, 3), dtype=torch.float32, device=dev)}
        if rgb0_p is not None:
            cols["rgb1"] = torch.empty((R, 3), dtype=torch.float32, device=dev)
        if w2 is not None:
            cols["rgb_awp"] = torch.empty((R, 3), dtype=torch.float32, device=dev)
    L.check(L.lib().evd_blur_loss_reduce(crf_rgb.handle, int(bool(skip_learn_crf)), L.ptr(rgb_p), L.ptr(rgb0_p), L.ptr(w1),
                                         L.ptr(w2), L.ptr(target), L.ptr(target_pts0), R, P, L.ptr(partial),
                                         L.ptr(cols.get("rgb")), L.ptr(cols.get("rgb1")), L.ptr(cols.get("rgb_awp")),
                                         L.stream_ptr()), "evd_blur_loss_reduce")
    return partial, cols


class _BlurLoss(torch.autograd.Function):
    """blur_loss_partials as an autograd node: forward = evd_blur_loss_reduce, backward = evd_blur_loss_bwd (gradients for the
    sub-exposure colours and for the RBK / AWP composition weights; identity / gamma response curve)."""

    @staticmethod
    def forward(ctx, crf_rgb, skip_learn, rgb_p, rgb0_p, w1, w2, target, target_pts0):
        partial, _ = blur_loss_partials(crf_rgb, rgb_p.detach(), w1.detach(), target, rgb0_p=None if rgb0_p is None else rgb0_p.detach(),
                                        w2=None if w2 is None else w2.detach(), target_pts0=target_pts0, skip_learn_crf=skip_learn)
        ctx.crf, ctx.skip = crf_rgb, bool(skip_learn)
        ctx.save_for_backward(*[t if t is not None else torch.empty(0, device=w1.device) for t in (rgb_p, rgb0_p, w1, w2, target, target_pts0)])
        ctx.has = [t is not None for t in (rgb_p, rgb0_p, w1, w2, target, target_pts0)]
        return partial

    @staticmethod
    def backward(ctx, g_partial):
        ts = [t.contiguous().float() if h else None for t, h in zip(ctx.saved_tensors, ctx.has)]
        rgb_p, rgb0_p, w1, w2, target, target_pts0 = ts
        R, P = w1.shape
        g = g_partial.detach().float().contiguous()              # dL/d partial stays on the device: no host copy in the backward
        d_rgb = torch.empty_like(rgb_p)
        d_rgb0 = torch.empty_like(rgb0_p) if rgb0_p is not None else None
        d_w1 = torch.empty_like(w1)
        d_w2 = torch.empty_like(w2) if w2 is not None else None
        L.check(L.lib().evd_blur_loss_bwd_dev(ctx.crf.handle, int(ctx.skip), L.ptr(rgb_p), L.ptr(rgb0_p), L.ptr(w1), L.ptr(w2), L.ptr(target),
                                              L.ptr(target_pts0), R, P, L.ptr(g), L.ptr(d_rgb), L.ptr(d_rgb0),
                                              L.ptr(d_w1), L.ptr(d_w2), L.stream_ptr()), "evd_blur_loss_bwd_dev")
        return None, None, d_rgb, d_rgb0, d_w1, d_w2, None, None


def blur_loss_partials_autograd(crf_rgb, rgb_p, w1, target, rgb0_p=None, w2=None, target_pts0=None, skip_learn_crf=False):
    """Differentiable form of blur_loss_partials (rgb_p [R,P,3], rgb0_p, w1 [R,P], w2 may require grad); returns the [8] partials."""
    c = lambda t: t.contiguous().float() if t is not None else None
    return _BlurLoss.apply(crf_rgb, skip_learn_crf, c(rgb_p), c(rgb0_p), c(w1), c(w2), c(target), c(target_pts0))


_COEF = {}


def _coef(device, values):
    """constant coefficient vector on the device, made once per (device, values)"""
    key = (str(device), tuple(float(v) for v in values))
    c = _COEF.get(key)
    if c is None:
        c = _COEF[key] = torch.tensor(key[1], dtype=torch.float32, device=device)
    return c


def blur_loss_from_partials(p, fine_loss_weight=None, w_pts0=0.0):
    """Assemble run_nerf.py:451-497 from the (all-reduced) partial vector. Returns (loss, dict of terms).
    loss = ((img + img0) (1 - flw) + fine flw + (pts0 + pts0_0) w_pts0) / n as ONE dot product with a constant coefficient vector (three
    launches forward, instead of a dozen 0-dim tensor operations and their backward nodes); the terms are detached views for logging."""
    flw = 0.0 if fine_loss_weight is None else float(fine_loss_weight)
    c = _coef(p.device, (1.0 - flw, 1.0 - flw, flw, float(w_pts0), float(w_pts0)))
    loss = (p[:5] * c).sum() / p.detach()[5]                 # the count is a constant of the batch
    t = p.detach()[:5] / p.detach()[5]
    terms = {"img_loss": t[0] + t[1], "pts0": t[3] + t[4]}
    if fine_loss_weight is not None:
        terms["img_fine_loss"] = t[2]
    return loss, terms


def event_loss_partials(crf_ev, start, end, cum_neg, cum_pos, thr_neg, thr_pos, start0=None, end0=None,
                        add_bii="pos-neg", tonemap_only=False, color_mask=None, color_weight=None,
                        skip_learn_crf=False, partial=None):
    dev = start.device
    if partial is None:
        partial = torch.zeros((EVENT_PARTIALS,), dtype=torch.float32, device=dev)
    c = lambda t: t.contiguous().float() if t is not None else None
    start, end, start0, end0, cum_neg, cum_pos = c(start), c(end), c(start0), c(end0), c(cum_neg), c(cum_pos)
    N = start.shape[0]
    cm = color_mask.contiguous().to(torch.uint8) if color_mask is not None else None
    cw = np.ascontiguousarray(color_weight, dtype=np.float32) if color_weight is not None else None
    mode = {None: 0, "none": 0, "pos-neg": 1, "color-pos-neg": 2}[add_bii]
    L.check(L.lib().evd_event_loss_reduce(crf_ev.handle, int(bool(skip_learn_crf)), mode, int(bool(tonemap_only)),
                                          L.ptr(start), L.ptr(end), L.ptr(start0), L.ptr(end0), L.ptr(cum_neg), L.ptr(cum_pos),
                                          float(thr_neg), float(thr_pos), L.ptr(cm),
                                          cw.ctypes.data_as(C.POINTER(C.c_float)) if cw is not None else None, N,
                                          L.ptr(partial), L.stream_ptr()), "evd_event_loss_reduce")
    return partial


class _EventLoss(torch.autograd.Function):
    """event_loss_partials as an autograd node: forward = evd_event_loss_reduce, backward = evd_event_loss_bwd (gradients for
    the four colour inputs and for the learnable event-CRF parameters, returned as ONE flat tensor in the layout of
    evd_event_loss_bwd; ``crf_param_grads`` splits it into the state-dict shapes)."""

    @staticmethod
    def forward(ctx, crf_ev, params, start, end, start0, end0, cum_neg, cum_pos, thr_neg, thr_pos, mode, tonemap_only, cm, cw, skip):
        partial = event_loss_partials(crf_ev, start.detach(), end.detach(), cum_neg, cum_pos, thr_neg, thr_pos,
                                      start0=None if start0 is None else start0.detach(), end0=None if end0 is None else end0.detach(),
                                      add_bii={0: None, 1: "pos-neg", 2: "color-pos-neg"}[mode], tonemap_only=tonemap_only, color_mask=cm,
                                      color_weight=cw, skip_learn_crf=skip)
        ctx.args = (crf_ev, float(thr_neg), float(thr_pos), int(mode), bool(tonemap_only), cw, bool(skip), start0 is not None)
        dev = start.device
        e = torch.empty(0, device=dev)
        ctx.save_for_backward(start, end, start0 if start0 is not None else e, end0 if end0 is not None else e, cum_neg, cum_pos,
                              cm if cm is not None else e)
        ctx.has_cm = cm is not None
        return partial

    @staticmethod
    def backward(ctx, g_partial):
        crf_ev, thr_neg, thr_pos, mode, tonemap_only, cw, skip, have0 = ctx.args
        start, end, start0, end0, cum_neg, cum_pos, cm = ctx.saved_tensors
        N = start.shape[0]
        g = g_partial.detach().float().contiguous()              # stays on the device
        d_start, d_end = torch.zeros_like(start), torch.zeros_like(end)
        d_start0 = torch.zeros_like(start0) if have0 else None
        d_end0 = torch.zeros_like(end0) if have0 else None
        d_params = torch.empty((int(L.lib().evd_crf_param_count()),), dtype=torch.float32, device=start.device)
        cwa = np.ascontiguousarray(cw, dtype=np.float32) if cw is not None else None
        L.check(L.lib().evd_event_loss_bwd_dev(crf_ev.handle, int(skip), mode, int(tonemap_only), L.ptr(start), L.ptr(end),
                                               L.ptr(start0) if have0 else None, L.ptr(end0) if have0 else None, L.ptr(cum_neg), L.ptr(cum_pos),
                                               thr_neg, thr_pos, L.ptr(cm) if ctx.has_cm else None,
                                               cwa.ctypes.data_as(C.POINTER(C.c_float)) if cwa is not None else None, N, L.ptr(g),
                                               L.ptr(d_start), L.ptr(d_end), L.ptr(d_start0), L.ptr(d_end0), L.ptr(d_params), L.stream_ptr()),
                "evd_event_loss_bwd_dev")
        return (None, d_params, d_start, d_end, d_start0, d_end0) + (None,) * 9


def event_loss_partials_autograd(crf_ev, crf_params, start, end, cum_neg, cum_pos, thr_neg, thr_pos, start0=None, end0=None,
                                 add_bii="pos-neg", tonemap_only=False, color_mask=None, color_weight=None, skip_learn_crf=False):
    """Differentiable form of event_loss_partials.  ``crf_params`` is a flat float32 leaf tensor of evd_crf_param_count()
    elements that stands for the event-CRF's parameters in the autograd graph (its .grad receives dL/d parameters in the layout
    of evd_event_loss_bwd; ``crf_param_grads`` maps that to the reference's state-dict names); the VALUES are the ones the
    ``crf_ev`` handle was created with."""
    c = lambda t: t.contiguous().float() if t is not None else None
    mode = {None: 0, "none": 0, "pos-neg": 1, "color-pos-neg": 2}[add_bii]
    cm = color_mask.contiguous().to(torch.uint8) if color_mask is not None else None
    return _EventLoss.apply(crf_ev, crf_params, c(start), c(end), c(start0), c(end0), c(cum_neg), c(cum_pos), thr_neg, thr_pos, mode,
                            tonemap_only, cm, color_weight, skip_learn_crf)


def crf_param_grads(flat, extra_features):
    """Split the flat parameter gradient of evd_event_loss_bwd into the reference's ``linear.{0,2,4,6}.{weight,bias}`` shapes."""
    nin, o = 1 + extra_features, 0
    out = {}
    out["linear.0.weight"] = flat[o:o + 128].reshape(16, 8)[:, :nin].clone(); o += 128
    out["linear.0.bias"] = flat[o:o + 16].clone(); o += 16
    out["linear.2.weight"] = flat[o:o + 256].reshape(16, 16).clone(); o += 256
    out["linear.2.bias"] = flat[o:o + 16].clone(); o += 16
    out["linear.4.weight"] = flat[o:o + 256].reshape(16, 16).clone(); o += 256
    out["linear.4.bias"] = flat[o:o + 16].clone(); o += 16
    out["linear.6.weight"] = flat[o:o + 16].reshape(1, 16).clone(); o += 16
    out["linear.6.bias"] = flat[o:o + 1].clone()
    return out


def event_loss_from_partials(p, stages=("stage0", "stage1")):
    """extra_loss['event_egm'] of run_nerf.py:559-572 from the (all-reduced) partials."""
    if "stage0" in stages and "stage1" in stages:
        return p[:2].sum() / p.detach()[2]                   # sum of the event weights: a constant of the batch
    if "stage0" in stages:
        return p[1] / p.detach()[2]
    if "stage1" in stages:
        return p[0] / p.detach()[2]
    return 0.0


def img2mse(x, y):
    """utils/metrics.py:7 through the fused reduction (identity CRF, P = 1)."""
    from .tonemapping import CRF
    xx = x.reshape(-1, 3)
    ones = torch.ones((xx.shape[0], 1), dtype=torch.float32, device=xx.device)
    p, _ = blur_loss_partials(_identity_crf(), xx, ones, y.reshape(-1, 3))
    return p[0] / p[5]


_ID_CRF = None


def _identity_crf():
    global _ID_CRF
    if _ID_CRF is None:
        from .tonemapping import CRF
        _ID_CRF = CRF("none")
    return _ID_CRF


def egm_loss(luma_start, luma_end, bii, color_mask=None, color_weight=None, log_eps=1e-5):
    """utils/events.py:260-284 on already tone-mapped lumas ([N,1] or [N,3] + one-hot mask)."""
    assert log_eps == 1e-5
    N = luma_start.shape[0]
    if luma_start.shape[-1] == 1:
        ls, le = luma_start.expand(N, 3), luma_end.expand(N, 3)
        cm = torch.zeros((N, 3), dtype=torch.uint8, device=ls.device)
        cm[:, 0] = 1
        cw = None
    else:
        ls, le, cm, cw = luma_start, luma_end, color_mask, color_weight
    zeros = torch.zeros((N,), dtype=torch.float32, device=ls.device)
    # bii enters as thr_neg * cum_neg with thr_neg = 1, cum_neg = bii
    p = event_loss_partials(_identity_crf(), ls, le, bii, zeros, 1.0, 0.0, add_bii=None, tonemap_only=True,
                            color_mask=cm, color_weight=cw)
    return p[0] / p[2]
