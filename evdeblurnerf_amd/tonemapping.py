"""Mirror of the reference ``networks/tonemapping.py`` (CRF :7-93, TonemappingTransform :96-154)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib as L

_LUMA = {"rec601": 0, "rec709": 1, "avg": 2}


def _np32(v):
    if isinstance(v, torch.Tensor):
        v = v.detach().cpu().numpy()
    return np.ascontiguousarray(v, dtype=np.float32)


class CRF:
    """map_type in {'none','gamma','learn'}; 'learn' takes the ``linear.{0,2,4,6}.{weight,bias}`` arrays."""

    def __init__(self, map_type: str, gamma: float = 2.2, state_dict=None, prefix="", extra_features=0):
        assert map_type in ("none", "gamma", "learn")
        self.map_type, self.gamma, self.extra_features = map_type, gamma, extra_features
        d = L.CrfDesc()
        d.map_type = {"none": 0, "gamma": 1, "learn": 2}[map_type]
        d.gamma, d.extra_features = float(gamma), int(extra_features)
        keep = []
        if map_type == "learn":
            if state_dict is None:
                raise L.EvdError("learn CRF needs its parameters (the reference's init_identity training is the "
                                 "caller's job: tonemapping.py:29-57)")
            for j, idx in enumerate((0, 2, 4, 6)):
                w, b = _np32(state_dict[f"{prefix}linear.{idx}.weight"]), _np32(state_dict[f"{prefix}linear.{idx}.bias"])
                keep += [w, b]
                d.w[j] = w.ctypes.data_as(C.POINTER(C.c_float))
                d.b[j] = b.ctypes.data_as(C.POINTER(C.c_float))
        h = C.c_void_p()
        L.check(L.lib().evd_crf_create(C.byref(d), C.byref(h)), "evd_crf_create")
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and L is not None and getattr(L, "lib", None) is not None:     # module globals may be gone at interpreter shutdown
            L.lib().evd_crf_destroy(h)
            self._h = None

    @property
    def handle(self):
        return self._h

    # training the learnable CRF: its parameters as one flat tensor in the layout of evd_event_loss_bwd (losses.crf_param_grads)
    def flat_params(self, device="cuda"):
        n = int(L.lib().evd_crf_param_count())
        host = np.zeros((n,), np.float32)
        L.check(L.lib().evd_crf_get_params(self._h, host.ctypes.data_as(C.c_void_p)), "evd_crf_get_params")
        return torch.tensor(host, device=device, requires_grad=True)

    def load_params(self, flat):
        """after optimizer.step(): 2.8 KB device -> host (the handle keeps the parameters on the host: one small synchronising copy)"""
        host = np.ascontiguousarray(flat.detach().cpu().numpy(), dtype=np.float32)
        L.check(L.lib().evd_crf_load_params(self._h, host.ctypes.data_as(C.c_void_p)), "evd_crf_load_params")

    def forward(self, x, x_feat=None, skip_learn=False, _luma=-1):
        sh = x.shape
        xx = x.reshape(-1, 3).contiguous().float()
        n = xx.shape[0]
        per_ch = 0
        ft = None
        if x_feat is not None and self.extra_features > 0:
            ft = x_feat.to(xx.dtype).contiguous()
            per_ch = int(ft.ndim == 3)
        out = torch.empty((n, 3 if _luma < 0 else 1), dtype=torch.float32, device=xx.device)
        L.check(L.lib().evd_crf_forward(self._h, L.ptr(xx), L.ptr(ft), per_ch, int(bool(skip_learn)), _luma, n,
                                        L.ptr(out), L.stream_ptr()), "evd_crf_forward")
        return out.reshape(*sh[:-1], out.shape[-1])

    __call__ = forward


class TonemappingTransform:
    def __init__(self, map_type_rgb: str, map_type_event: str, gamma: float = 2.2, luma_standard="rec601",
                 state_dict=None, extra_features_event=0, extra_features_rgb=0):
        assert luma_standard in _LUMA
        self.tonemapping_rgb = CRF(map_type_rgb, gamma, state_dict, "tonemapping_rgb.", extra_features_rgb)
        self.tonemapping_event = CRF(map_type_event, gamma, state_dict, "tonemapping_event.", extra_features_event)
        self.luma_standard = luma_standard

    def train(self, mode=True):
        return self

    def eval(self):
        return self

    def encode_rgb(self, x, skip_learn_crf=False, rgb_extra_feat=None, **kwargs):      # tonemapping.py:111-118
        assert x.shape[-1] == 3
        return self.tonemapping_rgb(x, skip_learn=skip_learn_crf, x_feat=rgb_extra_feat)

    def encode_luma(self, x, keep_rgb=False, tonemap_only=False, skip_learn_crf=False, ev_extra_feat=None, **kwargs):
        # tonemapping.py:120-139
        if tonemap_only:
            return self.tonemapping_event(x, skip_learn=skip_learn_crf, x_feat=ev_extra_feat)
        y = self.tonemapping_event.forward(x, skip_learn=skip_learn_crf, x_feat=ev_extra_feat, _luma=_LUMA[self.luma_standard])
        return torch.cat([y] * 3, dim=-1) if keep_rgb else y

    def forward(self, x, mode="encode", chunk=None, **kwargs):                          # tonemapping.py:141-154
        if x is None:
            return None
        if mode == "encode_rgb":
            return self.encode_rgb(x, **kwargs)
        if mode == "encode_luma":
            return self.encode_luma(x, **kwargs)
        raise RuntimeError(f"mode '{mode}' not recognized")

    __call__ = forward
