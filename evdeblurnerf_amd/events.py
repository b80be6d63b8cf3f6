"""Event preprocessing on the device (reference ``utils/events.py``): the per-pixel successor graph the event loader builds
once per dataset (``data/loader_events.py`` -> ``compute_successor``)."""
from __future__ import annotations

import torch

from . import _lib as L


def compute_successor(pixel_ids, num_pixels):
    """utils/events.py:72-120 with flat pixel ids (y * w + x, int), events in stream order.
    Returns (successor_idx int64 [N], num_successors int32 [N], latest_seen_idx int64 [num_pixels], first_seen_idx int64 [num_pixels])
    exactly as the reference's loop: latest_seen holds each pixel's FIRST event, first_seen its LAST one."""
    ids = pixel_ids.to(torch.int32).contiguous()
    dev = ids.device
    n = ids.shape[0]
    succ = torch.empty((n,), dtype=torch.int64, device=dev)
    nsucc = torch.empty((n,), dtype=torch.int32, device=dev)
    latest = torch.empty((num_pixels,), dtype=torch.int64, device=dev)
    first = torch.empty((num_pixels,), dtype=torch.int64, device=dev)
    need = int(L.lib().evd_compute_successor_workspace_bytes(n))
    ws = torch.empty((need,), dtype=torch.uint8, device=dev)
    L.check(L.lib().evd_compute_successor(L.ptr(ids), n, int(num_pixels), L.ptr(succ), L.ptr(nsucc), L.ptr(latest), L.ptr(first),
                                          L.ptr(ws), need, L.stream_ptr()), "evd_compute_successor")
    return succ, nsucc, latest, first
