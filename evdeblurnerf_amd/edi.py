"""Event double integral prior on the GPU: mirror of the reference ``utils/edi.py``
(brightness_increment_image :44-70 with bilinear splat :7-41, deblur_double_integral :91-95)."""
from __future__ import annotations

import torch

from . import _lib as L


def brightness_increment_image(x, y, p, w, h, c_pos, c_neg, interpolate=True, color_events=False):
    if color_events:
        raise NotImplementedError("Bayer demosaicing of colour events (cv2.cvtColor) is CPU data preparation")
    if not interpolate:
        x, y = torch.floor(x), torch.floor(y)
    x, y = x.contiguous().float(), y.contiguous().float()
    p = p.contiguous().to(torch.int8)
    img = torch.empty((h, w), dtype=torch.float32, device=x.device)
    L.check(L.lib().evd_edi_bii_image(L.ptr(x), L.ptr(y), L.ptr(p), x.shape[0], int(w), int(h), float(c_pos), float(c_neg),
                                      L.ptr(img), L.stream_ptr()), "evd_edi_bii_image")
    return img


def deblur_double_integral(blurry, bii):
    """blurry [H,W(,3)], bii [steps-1, H,W(,3)] -> sharp, utils/edi.py:91-95."""
    b = blurry.contiguous().float()
    e = bii.contiguous().float()
    out = torch.empty_like(b)
    L.check(L.lib().evd_edi_deblur(L.ptr(b), L.ptr(e), e.shape[0] + 1, b.numel(), L.ptr(out), L.stream_ptr()), "evd_edi_deblur")
    return out
