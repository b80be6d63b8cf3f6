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

    def __init__(self, map_type: str, gamma: float = 2.2, state_dict=None, prefix="", extra_features=0, init_identity=False,
                 init_seed=42):
        assert map_type in ("none", "gamma", "learn")
        self.map_type, self.gamma, self.extra_features = map_type, gamma, extra_features
        if map_type == "learn" and state_dict is None and init_identity:
            state_dict, prefix = self.identity_state_dict(extra_features, init_seed), ""
        d = L.CrfDesc()
        d.map_type = {"none": 0, "gamma": 1, "learn": 2}[map_type]
        d.gamma, d.extra_features = float(gamma), int(extra_features)
        keep = []
        if map_type == "learn":
            if state_dict is None:
                raise L.EvdError("learn CRF needs its parameters: pass state_dict or init_identity=True (tonemapping.py:29-57)")
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
        self._take_pending()
        return self._h

    @staticmethod
    def identity_state_dict(extra_features=0, seed=42, steps=3000, device="cuda"):
        """CRF.init_identity (tonemapping.py:29-57; `tone_mapping_learn_init_identity = True` in the shipped configs, wired at
        run_nerf.py:239): a freshly initialised 1+F -> 16 -> 16 -> 16 -> 1 MLP is pre-trained so that sigmoid(0.1 mlp([x, 0]) + x)
        reproduces x -- 3000 Adam steps (lr 1e-2) on batches of 64 x 3 uniform samples from a generator seeded with 42, the extra
        features held at zero.  A one-off initialiser of 625 parameters, run as the reference runs it (torch ops on the device);
        the result enters the library through evd_crf_create like any other state dict.  The reference's own draw (torch 1.13
        CUDA generator + nn.Linear default init) is not reproducible bit for bit: the PROCEDURE is the contract."""
        from torch import nn
        gen = torch.Generator(device=device).manual_seed(int(seed))
        cpu_gen = torch.Generator().manual_seed(int(seed))
        dims = [(1 + extra_features, 16), (16, 16), (16, 16), (16, 1)]
        layers = []
        for fi, fo in dims:
            lin = nn.Linear(fi, fo)
            bound = 1.0 / np.sqrt(fi)                    # nn.Linear's default: kaiming_uniform(a=sqrt(5)) = U(+-1/sqrt(fan_in))
            with torch.no_grad():
                lin.weight.copy_((torch.rand((fo, fi), generator=cpu_gen) * 2 - 1) * bound)
                lin.bias.copy_((torch.rand((fo,), generator=cpu_gen) * 2 - 1) * bound)
            layers += [lin, nn.ReLU()]
        mlp = nn.Sequential(*layers[:-1]).to(device)
        optim = torch.optim.Adam(mlp.parameters(), lr=1e-2)
        for _ in range(steps):
            x = torch.rand((64, 3), generator=gen, device=device)
            x_in = x.reshape(-1, 1)
            x_feat = torch.cat([x_in, torch.zeros((x_in.shape[0], extra_features), device=device)], -1) if extra_features > 0 else x_in
            y = torch.sigmoid(mlp(x_feat) * 0.1 + x_in).reshape(x.shape)
            loss = torch.mean((y - x) ** 2)
            optim.zero_grad()
            loss.backward()
            optim.step()
        return {f"linear.{i}.{k}": v.detach().cpu().numpy().astype(np.float32) for i in (0, 2, 4, 6)
                for k, v in (("weight", mlp[i].weight), ("bias", mlp[i].bias))}

    # training the learnable CRF: its parameters as one flat tensor in the layout of evd_event_loss_bwd (losses.crf_param_grads)
    def flat_params(self, device="cuda"):
        n = int(L.lib().evd_crf_param_count())
        host = np.zeros((n,), np.float32)
        L.check(L.lib().evd_crf_get_params(self.handle, host.ctypes.data_as(C.c_void_p)), "evd_crf_get_params")
        return torch.tensor(host, device=device, requires_grad=True)

    def load_params(self, flat, blocking=False):
        """after optimizer.step(): 2.8 KB device -> host (the handle keeps the parameters on the host: they travel as a kernel argument).
        The copy is queued into pinned memory and the handle takes the values at its NEXT use (``handle``), when the copy has long
        completed: the training loop's stream is not drained after every optimizer step.  blocking=True copies at once."""
        if blocking or not flat.is_cuda:
            host = np.ascontiguousarray(flat.detach().cpu().numpy(), dtype=np.float32)
            L.check(L.lib().evd_crf_load_params(self._h, host.ctypes.data_as(C.c_void_p)), "evd_crf_load_params")
            self._pending = None
            return
        if getattr(self, "_pin", None) is None or self._pin.numel() != flat.numel():
            self._pin = torch.empty(flat.numel(), dtype=torch.float32, pin_memory=True)
            self._pin_ev = torch.cuda.Event()
        self._pin.copy_(flat.detach().reshape(-1), non_blocking=True)
        self._pin_ev.record()
        self._pending = True

    def _take_pending(self):
        if getattr(self, "_pending", None):
            self._pin_ev.synchronize()
            L.check(L.lib().evd_crf_load_params(self._h, C.c_void_p(self._pin.data_ptr())), "evd_crf_load_params")
            self._pending = None

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
        L.check(L.lib().evd_crf_forward(self.handle, L.ptr(xx), L.ptr(ft), per_ch, int(bool(skip_learn)), _luma, n,
                                        L.ptr(out), L.stream_ptr()), "evd_crf_forward")
        return out.reshape(*sh[:-1], out.shape[-1])

    __call__ = forward


class TonemappingTransform:
    def __init__(self, map_type_rgb: str, map_type_event: str, gamma: float = 2.2, luma_standard="rec601",
                 init_learn_identity=False, extra_features_event=0, extra_features_rgb=0, state_dict=None):
        """tonemapping.py:98-109 (same positional order); `state_dict` (crf_state_dict of a checkpoint, run_nerf.py:631) is the
        extra way in for trained parameters; without it a 'learn' CRF needs init_learn_identity=True"""
        assert luma_standard in _LUMA
        self.tonemapping_rgb = CRF(map_type_rgb, gamma, state_dict, "tonemapping_rgb.", extra_features_rgb, init_identity=init_learn_identity)
        self.tonemapping_event = CRF(map_type_event, gamma, state_dict, "tonemapping_event.", extra_features_event,
                                     init_identity=init_learn_identity)
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
