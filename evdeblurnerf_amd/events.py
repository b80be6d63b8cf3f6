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
        if id_to_color_map is not None and not isinstance(id_to_color_map, torch.Tensor):
            id_to_color_map = torch.as_tensor(np.asarray(id_to_color_map))
        self.id_to_color_map = None if id_to_color_map is None else id_to_color_map.to(device=dev, dtype=torch.uint8).contiguous()
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
        head = (L.ptr(self.events), self.events.shape[0], self.events.shape[1], L.ptr(self.id_to_coords), self.id_to_coords.shape[0], L.ptr(self.id_to_color_map))
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


class EventTables:
    """The once-per-dataset half of the reference's event loader -- ``LLFFEventsDataset.load_event_data`` (data/loader_events.py:150-257)
    and ``load_events_h5`` (utils/events.py:11-69) -- on the arrays the files hold, computed on the device.  Reading ``events.h5`` /
    ``*.npy`` / ``*.npz`` stays with the caller (on-disk formats are out of scope; h5py is not in this image): pass the arrays.

    Attributes, under the keys of the reference's ``retvals``: ``events`` [N', 4] float64 (coordinate id, timestamp in microseconds, polarity
    -1 / 1, successor index), ``id_to_coords`` [Ncoords, 2] float64, ``coords_to_id`` ([h, w] int32 table for integer coordinates, else
    None: ``id_to_coords`` is in the byte order the ids were assigned in and is searched directly), ``id_to_color_map`` [Ncoords, 3] bool or
    None, ``events_num_successors`` [N'] int32, ``events_with_successor_idx`` int64, ``intcoords``, ``noev_coord_ids`` (the ids of the pixels
    no event rounds to), ``allknown_poses`` [M, 3, 4], ``allknown_poses_timestamps``, ``images_*`` timestamps, ``pose_track`` (poses.PoseTrack: the
    reference's ``events_pose_bspl`` inside ``interpolate_poses``).  ``sampler(K)`` returns the ready ``EventSampler``."""

    @classmethod
    def from_arrays(cls, x, y, t, p, h, w, all_timestamps, all_poses_bounds, img_timestamps=None, img_timestamps_start=None, img_timestamps_end=None,
                    ev_map=None, color_events=False, events_tms_unit="ns", events_tms_files_unit="us", event_accumulate_step_range=(0, 0),
                    event_accumulate_step_range_end=(0, 0), bd_scale=1.0, recenter=True, recenter_partial=None, device="cuda", check=True):
        """x, y, t, p: the four datasets of events.h5 (any numeric dtypes; t in ``events_tms_unit``); all_timestamps [M] and the
        img_timestamps* in ``events_tms_files_unit`` (all_timestamps.npy, images_1/timestamps.npz); all_poses_bounds [M, 17]
        (all_poses_bounds.npy); ev_map: None, or (inv_mapx, inv_mapy) [h, w] (ev_map.npz: rectified float coordinates).  check=True reads the
        two device flags back and raises where the reference asserts (polarities not {-1, 1}, a coordinate without colour)."""
        self = cls()
        dev = torch.device(device)
        powers = {"s": 0, "ms": -3, "us": -6, "ns": -9}                       # utils/misc.py:108-110 convert_unit(unit, "us")
        ev_scale = 10 ** (powers[events_tms_unit] - powers["us"])
        file_scale = 10 ** (powers[events_tms_files_unit] - powers["us"])
        key_t = np.asarray(all_timestamps).astype(np.float64) * file_scale     # :158-163 (the integer dtype games of possibly_smallest_int do not change values)
        apb = np.asarray(all_poses_bounds, dtype=np.float64)
        self.allknown_poses = apb[:, :-2].reshape(-1, 3, 5)[:, :3, :4].copy()  # :170-172
        self.allknown_poses_timestamps = key_t
        sc = lambda a: None if a is None else np.asarray(a, dtype=np.float64) * file_scale
        self.images_poses_timestamps, self.images_timestamps_start, self.images_timestamps_end = sc(img_timestamps), sc(img_timestamps_start), sc(img_timestamps_end)
        xs = torch.as_tensor(np.asarray(x), device=dev).to(torch.float32).contiguous()          # utils/events.py:35-36
        ys = torch.as_tensor(np.asarray(y), device=dev).to(torch.float32).contiguous()
        ts = (torch.as_tensor(np.asarray(t), device=dev).to(torch.float64) * ev_scale).contiguous()
        ps = torch.as_tensor(np.asarray(p), device=dev).to(torch.float64).contiguous()
        N, h, w = int(xs.shape[0]), int(h), int(w)
        lib = L.lib()
        # ---- coordinate ids (utils/events.py:39-66)
        ev_ids = torch.empty((max(N, 1),), dtype=torch.int64, device=dev)
        noev = torch.empty((h * w,), dtype=torch.int64, device=dev)
        i2c = torch.empty((N + h * w, 2), dtype=torch.float64, device=dev)
        counts = torch.zeros((2,), dtype=torch.int64, device=dev)
        nb = int(lib.evd_event_coord_ids_workspace_bytes(N, h, w))
        ws = torch.empty((nb,), dtype=torch.uint8, device=dev)
        L.check(lib.evd_event_coord_ids(L.ptr(xs), L.ptr(ys), N, h, w, L.ptr(ev_ids), L.ptr(noev), L.ptr(i2c), L.ptr(counts), L.ptr(ws), nb, L.stream_ptr()),
                "evd_event_coord_ids")
        n_coords, n_noev = (int(v) for v in counts.tolist())                  # (the one host read-back of the data-dependent sizes)
        self.id_to_coords = i2c[:n_coords].clone()
        self.noev_coord_ids = noev[:n_noev].clone()
        # ---- events inside the range of the known poses, polarity in {-1, 1} (:191, 203-206)
        ev3 = torch.empty((max(N, 1), 3), dtype=torch.float64, device=dev)
        cnt = torch.zeros((1,), dtype=torch.int64, device=dev)
        bad = torch.zeros((2,), dtype=torch.int32, device=dev)
        nb = int(lib.evd_event_filter_workspace_bytes(N))
        ws = torch.empty((nb,), dtype=torch.uint8, device=dev)
        L.check(lib.evd_event_filter(L.ptr(ev_ids), L.ptr(ts), L.ptr(ps), N, float(key_t.min()), float(key_t.max()), L.ptr(ev3), L.ptr(cnt), L.ptr(bad[0:1]),
                                     L.ptr(ws), nb, L.stream_ptr()), "evd_event_filter")
        n_ev = int(cnt.item())
        ev3 = ev3[:n_ev]
        self.intcoords = bool(torch.all(self.id_to_coords == torch.trunc(self.id_to_coords)).item()) if n_coords else True       # :194
        self.coords_to_id = None
        if self.intcoords:                                                     # :195-197
            c2i = torch.full((h, w), -1, dtype=torch.int32, device=dev)
            c2i[self.id_to_coords[:, 1].long(), self.id_to_coords[:, 0].long()] = torch.arange(n_coords, dtype=torch.int32, device=dev)
            self.coords_to_id = c2i
        # ---- colour of a coordinate id (:208-236)
        self.id_to_color_map = None
        if color_events:
            if self.intcoords and ev_map is not None:
                raise L.EvdError("Int coordinates but an ev_map was given. Are coordinates rectified?")          # (:216-217)
            if not self.intcoords and ev_map is None:
                raise L.EvdError("Float coordinates but no ev_map given. Are coordinates not rectified?")      # (:221-222)
            cmap = torch.zeros((max(n_coords, 1), 3), dtype=torch.uint8, device=dev)
            mx = my = None
            if ev_map is not None:
                mx = torch.as_tensor(np.asarray(ev_map[0]), device=dev).to(torch.float32).contiguous()
                my = torch.as_tensor(np.asarray(ev_map[1]), device=dev).to(torch.float32).contiguous()
                if tuple(mx.shape) != (h, w) or tuple(my.shape) != (h, w):
                    raise L.EvdError("ev_map: inv_mapx / inv_mapy [h, w]")
            nb = int(lib.evd_event_color_map_workspace_bytes(n_coords))
            ws = torch.empty((nb,), dtype=torch.uint8, device=dev)
            L.check(lib.evd_event_color_map(L.ptr(self.id_to_coords), n_coords, h, w, L.ptr(mx), L.ptr(my), L.ptr(self.noev_coord_ids), n_noev, L.ptr(cmap),
                                            L.ptr(bad[1:2]), L.ptr(ws), nb, L.stream_ptr()), "evd_event_color_map")
            self.id_to_color_map = cmap[:n_coords].bool()
        # ---- successor graph (:239), the augmented table (:245), the events a batch may start from (:248-255)
        succ, nsucc, _, _ = compute_successor(ev3[:, 0].to(torch.int32), max(n_coords, 1))
        self.events = torch.cat([ev3, succ[:n_ev, None].to(torch.float64)], dim=1)
        self.events_num_successors = nsucc[:n_ev]
        lo, hi = tuple(event_accumulate_step_range), tuple(event_accumulate_step_range_end)
        min_step = max(lo[0], hi[0]) if lo != (0, 0) else 0
        self.events_with_successor_idx = torch.nonzero(self.events_num_successors > min_step).reshape(-1)
        if check:
            flags = bad.tolist()
            if flags[0]:
                raise L.EvdError("polarities must be {0, 1} or {-1, 1} (loader_events.py:203-206)")
            if flags[1]:
                raise L.EvdError("a coordinate that carries events has no entry in the ev_map inverse maps (loader_events.py:231-234)")
        from .poses import PoseTrack
        self.pose_track = None
        if recenter_partial is not None or not recenter:
            self.pose_track = PoseTrack(key_t, self.allknown_poses, bd_scale=bd_scale, recenter=recenter, recenter_partial=recenter_partial, device=device)
        self.h, self.w, self.device = h, w, dev
        self._hops = (lo, hi)
        return self

    def interpolate_poses(self, t):
        """LLFFEventsDataset.interpolate_poses (:133-148) = events_pose_bspl (:175-182: Slerp + cubic spline of the known poses, clipped to
        their time range) + the LLFF column change, bd_scale and recentring: float32 [n, 4, 4] on the device"""
        if self.pose_track is None:
            raise L.EvdError("EventTables.interpolate_poses: built with recenter=True but without the image dataset's recenter_partial")
        return self.pose_track.interpolate_poses(t)

    def sampler(self, K, step_end=0, scheduler="constant"):
        """-> the EventSampler of this dataset (integer_coords, colour map, pose track, hop schedule when an accumulation range was given)"""
        if self.pose_track is None:
            raise L.EvdError("EventTables.sampler: built with recenter=True but without the image dataset's recenter_partial")
        s = EventSampler(self.events, self.id_to_coords.to(torch.float32), K=K, id_to_color_map=self.id_to_color_map, integer_coords=self.intcoords,
                         device=str(self.device), pose_track=self.pose_track)
        if self._hops[0] != (0, 0):
            s.set_hop_schedule(self.events_num_successors, self._hops[0], self._hops[1], step_end, scheduler)
        return s
