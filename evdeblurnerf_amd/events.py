"""Event preprocessing on the device (reference ``utils/events.py``): the per-pixel successor graph the event loader builds
once per dataset (``data/loader_events.py`` -> ``compute_successor``), and the per-iteration event batch assembly
(``EventsDataset.sample_events``, data/loader_events.py:259-304) on resident tables."""
from __future__ import annotations

import ctypes as C

import numpy as np
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


def annealing_interpolator(start_value, end_value, end_step, method="linear", start_step=0):
    """utils/misc.py:15-55 restated: step -> a value between start_value and end_value ('linear', 'cosine', 'constant'); the reference's
    linear branch multiplies the slope by `step`, not by `step - start_step` (:38) -- kept."""
    import math
    if method == "linear":
        def f(step):
            if step >= end_step:
                return end_value
            if step < start_step:
                return start_value
            return start_value + (end_value - start_value) / (end_step - start_step) * step
        return f
    if method == "cosine":
        def f(step):
            if step >= end_step:
                return end_value
            if step < start_step:
                return start_value
            c = (1 + math.cos(math.pi * (step - start_step) / (end_step - start_step))) / 2
            return start_value * c + end_value * (1 - c)
        return f
    if method == "constant":
        return lambda step: start_value
    raise ValueError("Unsupported method: {}".format(method))


def draw_hops(num_successors, min_step, max_step):
    """loader_events.py:262-268 + utils/misc.py:87-92 (torch_randint_vec): per event a uniform draw in [min_step - 1,
    min(max_step, num_successors) - 1 + 1e-5], rounded to int64 -- the same torch calls in the same order, so a seeded generator gives the
    reference's hops.  num_successors [n] (of the batch's events), on the device the draw should run on."""
    dev = num_successors.device
    mins = torch.tensor(min_step, device=dev) - 1
    maxs = torch.minimum(torch.tensor(max_step, device=dev), num_successors) - 1 + 1e-5
    values = torch.distributions.uniform.Uniform(mins.float(), maxs.float()).sample()
    return torch.round(values).to(torch.int64)


class EventSampler:
    """The tables EventsDataset keeps on the device (data/loader_events.py:60-75: events [N, ncol] float64 with the successor index in the
    last column, id_to_coords, id_to_color_map) and ``sample_events`` on them.

    The start / end poses: the reference interpolates them per batch on the CPU (scipy Slerp + cubic spline through
    ``interpolate_poses(timestamps.cpu().numpy())``, :280-283).  With ``pose_track`` (poses.PoseTrack) the same interpolation runs
    inside the batch kernel at the events' own timestamps; with ``poses`` ([N, 3, 4] float32, 48 bytes per event) the caller supplies
    one pose per event, evaluated once per dataset (a pose is a pure function of the timestamp)."""

    def __init__(self, events, id_to_coords, poses=None, K=None, id_to_color_map=None, integer_coords=True, device="cuda", pose_track=None):
        dev = torch.device(device)
        if K is None:
            raise L.EvdError("EventSampler: K is required")
        if (poses is None) == (pose_track is None):
            raise L.EvdError("EventSampler: exactly one of poses ([N, 3, 4]) and pose_track (poses.PoseTrack)")
        self.events = torch.as_tensor(events, dtype=torch.float64, device=dev).contiguous()
        self.id_to_coords = torch.as_tensor(id_to_coords, dtype=torch.float32, device=dev).contiguous()
        if self.events.dim() != 2 or self.events.shape[1] < 4:
            raise L.EvdError("EventSampler: events [N, >= 4] float64")
        self.pose_track = pose_track
        self.poses = None
        if poses is not None:
            self.poses = torch.as_tensor(poses, dtype=torch.float32, device=dev).reshape(-1, 3, 4).contiguous()
            if self.poses.shape[0] != self.events.shape[0]:
                raise L.EvdError("EventSampler: one pose [3, 4] per event")
        self.id_to_color_map = None if id_to_color_map is None else torch.as_tensor(np.asarray(id_to_color_map), device=dev).to(torch.uint8).contiguous()
        self.K = np.ascontiguousarray(np.asarray(K, dtype=np.float32).reshape(-1))
        self.integer_coords = bool(integer_coords)
        self._mismatch = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.num_successors = None
        self._hop_schedule = None

    def set_hop_schedule(self, events_num_successors, step_range, step_range_end, step_end, scheduler="linear"):
        """loader_events.py:53-70: the tables / schedules behind the multi-hop branch of sample_events (event_accumulate_step_range,
        _range_end, _step_end, _step_scheduler).  After this, sample_events(events_ids, global_step=...) draws the hops as the reference."""
        self.num_successors = torch.as_tensor(events_num_successors, device=self.events.device)
        self._hop_schedule = (annealing_interpolator(step_range[0], step_range_end[0], step_end, scheduler),
                              annealing_interpolator(step_range[1], step_range_end[1], step_end, scheduler))

    def hops_for(self, events_ids, global_step):
        """-> hops [n] int64, or None in the single-hop branch ((min_step, max_step) == (0, 0), loader_events.py:262-264)"""
        if self._hop_schedule is None:
            return None
        min_step, max_step = int(self._hop_schedule[0](global_step)), int(self._hop_schedule[1](global_step))
        if (min_step, max_step) == (0, 0):
            return None
        ids = events_ids.to(device=self.events.device, dtype=torch.int64)
        return draw_hops(self.num_successors[ids], min_step, max_step)

    def interpolate_poses(self, t):
        """loader_events.py:133-148 (needs pose_track)."""
        if self.pose_track is None:
            raise L.EvdError("EventSampler.interpolate_poses: constructed with a pose table, not a pose_track")
        return self.pose_track.interpolate_poses(t)

    def sample_events(self, events_ids, hops=None, check=False, global_step=None):
        """loader_events.py:259-304.  hops None: the branch of the shipped configs (event_accumulate_step_range [0, 0]); else the number of
        successor hops per event (the reference draws them, :265-268; gather_successor follows hops + 1 links).  Returns the reference's
        dict (rays [n, 3, 2], polarity sums float32).  check=True reads the device flag back (the reference's assert :284; also set by an
        id outside the table or, in the single-hop branch, an event without successor)."""
        if hops is None and global_step is not None:          # the reference's signature: sample_events(events_ids, global_step)
            hops = self.hops_for(events_ids, global_step)
        ids = events_ids.to(device=self.events.device, dtype=torch.int64).contiguous()
        n, dev = ids.shape[0], self.events.device
        hp = hops.to(device=dev, dtype=torch.int64).contiguous() if hops is not None else None
        f32 = dict(dtype=torch.float32, device=dev)
        rs, re = torch.empty((n, 3, 2), **f32), torch.empty((n, 3, 2), **f32)
        pos, neg = torch.empty((n,), **f32), torch.empty((n,), **f32)
        cid = torch.empty((n,), dtype=torch.int64, device=dev)
        cm = torch.empty((n, 3), dtype=torch.uint8, device=dev) if self.id_to_color_map is not None else None
        head = (L.ptr(self.events), self.events.shape[0], self.events.shape[1], L.ptr(self.id_to_coords), L.ptr(self.id_to_color_map))
        tail = (L.ptr(ids), L.ptr(hp), n, self.K.ctypes.data_as(C.POINTER(C.c_float)), int(self.integer_coords),
                L.ptr(rs), L.ptr(re), L.ptr(pos), L.ptr(neg), L.ptr(cid), L.ptr(cm), None, L.ptr(self._mismatch), L.stream_ptr())
        if self.pose_track is not None:
            L.check(L.lib().evd_sample_events_track(*head, C.byref(self.pose_track.struct), *tail), "evd_sample_events_track")
        else:
            L.check(L.lib().evd_sample_events(*head, L.ptr(self.poses), *tail), "evd_sample_events")
        if check and int(self._mismatch.item()):
            raise L.EvdError("sample_events: an id outside the event table, an event without successor, or an end event that is not on "
                             "its start event's coordinate id")
        return {"events_pos_pol_cumsum": pos, "events_neg_pol_cumsum": neg, "events_rays_start": rs, "events_rays_end": re,
                "events_coords_ids": cid, "events_color_map": cm.bool() if cm is not None else None}
