"""Camera trajectory on the device: the reference's ``LLFFEventsDataset.interpolate_poses`` (data/loader_events.py:133-148) on top of
``utils/data.py:34-62 _get_slerp_interpolator`` (scipy ``Slerp`` of the key rotations + cubic ``interp1d`` of the key translations).

The reference evaluates it on the CPU for every event batch.  Here the part that does not depend on the query is prepared ONCE on the
host in plain numpy (no scipy at run time): key rotations as unit quaternions, the rotation vectors between neighbours, and the
not-a-knot cubic spline as one polynomial per interval; ``evd_interpolate_poses`` / ``evd_sample_events_track`` evaluate a pose per
timestamp on the device (csrc/pose_track.h)."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib as L


def _orthogonalize(M):
    """scipy >= 1.10 Rotation.from_matrix: the nearest rotation of each (nearly orthogonal) matrix, U @ Vt of its SVD."""
    U, _, Vt = np.linalg.svd(M)
    return U @ Vt


def quat_from_matrix(M):
    """Rotation.from_matrix's conversion (largest of the diagonal / trace decides the branch), normalised; [n, 4] as (x, y, z, w)."""
    M = np.asarray(M, np.float64)
    q = np.empty((M.shape[0], 4))
    for n_ in range(M.shape[0]):
        m = M[n_]
        dec = (m[0, 0], m[1, 1], m[2, 2], m[0, 0] + m[1, 1] + m[2, 2])
        c = int(np.argmax(dec))
        if c != 3:
            i, j, k = c, (c + 1) % 3, (c + 2) % 3
            q[n_, i] = 1 - dec[3] + 2 * m[i, i]
            q[n_, j] = m[j, i] + m[i, j]
            q[n_, k] = m[k, i] + m[i, k]
            q[n_, 3] = m[k, j] - m[j, k]
        else:
            q[n_, 0] = m[2, 1] - m[1, 2]
            q[n_, 1] = m[0, 2] - m[2, 0]
            q[n_, 2] = m[1, 0] - m[0, 1]
            q[n_, 3] = 1 + dec[3]
        q[n_] /= np.linalg.norm(q[n_])
    return q


def _qmul(p, q):
    px, py, pz, pw = p[..., 0], p[..., 1], p[..., 2], p[..., 3]
    qx, qy, qz, qw = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    return np.stack([pw * qx + px * qw + py * qz - pz * qy, pw * qy - px * qz + py * qw + pz * qx,
                     pw * qz + px * qy - py * qx + pz * qw, pw * qw - px * qx - py * qy - pz * qz], -1)


def _as_rotvec(q):
    """Rotation.as_rotvec: w >= 0 (the short way round), angle = 2 atan2(|v|, w), series below 1e-3."""
    q = np.where(q[..., 3:] < 0, -q, q)
    nv = np.linalg.norm(q[..., :3], axis=-1)
    ang = 2 * np.arctan2(nv, q[..., 3])
    small = ang <= 1e-3
    sc = np.where(small, 2 + ang ** 2 / 12 + 7 * ang ** 4 / 2880, ang / np.where(small, 1.0, np.sin(ang / 2)))
    return q[..., :3] * sc[..., None]


def notaknot_cubic(x, y):
    """interp1d(kind='cubic') = make_interp_spline(k=3) with not-a-knot ends, as one cubic per interval in u = (t - x_i) / h_i:
    returns [M-1, 4, dim] (c0..c3).  Second derivatives from the tridiagonal system with the two end unknowns eliminated by the
    not-a-knot conditions (third derivative continuous at x_1 and x_M-2)."""
    x = np.asarray(x, np.float64)
    y = np.asarray(y, np.float64)
    M = x.shape[0]
    if M < 4:
        raise L.EvdError("the cubic translation spline needs >= 4 key poses (scipy interp1d(kind='cubic'))")
    h = np.diff(x)
    if not np.all(h > 0):
        raise L.EvdError("key timestamps must be strictly ascending")
    d = np.diff(y, axis=0) / h[:, None]
    n = M - 2
    a = h[:-1].copy()
    b = 2 * (h[:-1] + h[1:])
    c = h[1:].copy()
    r = 6 * (d[1:] - d[:-1])
    b[0] += h[0] * (1 + h[0] / h[1])
    c[0] -= h[0] * h[0] / h[1]
    b[-1] += h[-1] * (1 + h[-1] / h[-2])
    a[-1] -= h[-1] * h[-1] / h[-2]
    cp = np.zeros(n)
    rp = np.zeros_like(r)
    cp[0] = c[0] / b[0]
    rp[0] = r[0] / b[0]
    for i in range(1, n):
        den = b[i] - a[i] * cp[i - 1]
        cp[i] = c[i] / den
        rp[i] = (r[i] - a[i] * rp[i - 1]) / den
    m = np.zeros((M, y.shape[1]))
    m[n] = rp[n - 1]
    for i in range(n - 2, -1, -1):
        m[i + 1] = rp[i] - cp[i] * m[i + 2]
    m[0] = (1 + h[0] / h[1]) * m[1] - (h[0] / h[1]) * m[2]
    m[-1] = (1 + h[-1] / h[-2]) * m[-2] - (h[-1] / h[-2]) * m[-3]
    hh = h[:, None]
    return np.stack([y[:-1], hh * d - hh * hh * (2 * m[:-1] + m[1:]) / 6, hh * hh * m[:-1] / 2, hh * hh * (m[1:] - m[:-1]) / 6], 1)


class PoseTrack:
    """What LLFFEventsDataset keeps for its pose queries (loader_events.py:167-182: all_timestamps, all_poses [M, 3, 4] from
    all_poses_bounds.npy) + the transformation of the image dataset it re-applies (bd_scale, recenter_partial: run_nerf.py:74-81).

    ``interpolate_poses(t)`` mirrors the method of that name; ``EventSampler(pose_track=...)`` evaluates the same poses inside the
    event batch kernel."""

    def __init__(self, timestamps, poses, bd_scale=1.0, recenter=True, recenter_partial=None, spherify=False, device="cuda"):
        if spherify:
            raise NotImplementedError("spherify: no shipped config sets --spherify, and the reference's event path returns float64 "
                                      "poses there (utils/data.py:189-252)")
        ts = np.ascontiguousarray(timestamps, dtype=np.float64).reshape(-1)
        P = np.asarray(poses, dtype=np.float64)[:, :3, :4]
        if P.shape[0] != ts.shape[0]:
            raise L.EvdError("PoseTrack: one [3, 4] pose per timestamp")
        q = quat_from_matrix(_orthogonalize(P[:, :, :3]))
        qinv = q[:-1] * np.array([-1.0, -1.0, -1.0, 1.0])
        rotvec = _as_rotvec(_qmul(qinv, q[1:]))
        coef = notaknot_cubic(ts, P[:, :, 3])
        dev = torch.device(device)
        self.device = dev
        self.n_keys = int(ts.shape[0])
        self._t = torch.as_tensor(ts, device=dev)
        self._q = torch.as_tensor(np.ascontiguousarray(q), device=dev)
        self._rv = torch.as_tensor(np.ascontiguousarray(rotvec), device=dev)
        self._cf = torch.as_tensor(np.ascontiguousarray(coef), device=dev)
        self.bd_scale = float(bd_scale)
        self.recenter = bool(recenter)
        st = L.PoseTrack()
        st.n_keys = self.n_keys
        st.key_t, st.key_quat, st.key_rotvec, st.trans_coef = (self._t.data_ptr(), self._q.data_ptr(), self._rv.data_ptr(), self._cf.data_ptr())
        st.bd_scale = self.bd_scale
        st.recenter = int(self.recenter)
        if self.recenter:
            if recenter_partial is None:
                raise L.EvdError("PoseTrack: recenter=True needs the image dataset's recenter_partial (its average pose, run_nerf.py:80)")
            c2w = np.asarray(recenter_partial, dtype=np.float64)
            c44 = np.concatenate([c2w[:3, :4], np.array([[0, 0, 0, 1.0]])], 0)
            inv = np.linalg.inv(c44)
            for i in range(12):
                st.recenter_inv[i] = float(inv[i // 4, i % 4])
        self.struct = st

    def interpolate_poses(self, t):
        """loader_events.py:133-148 -> float32 [n, 4, 4] on the device (t: array or tensor of timestamps, any float / int dtype)."""
        tt = torch.as_tensor(t).to(device=self.device, dtype=torch.float64).contiguous().reshape(-1)
        n = tt.shape[0]
        out = torch.empty((n, 4, 4), dtype=torch.float32, device=self.device)
        p34 = torch.empty((n, 3, 4), dtype=torch.float32, device=self.device)
        import ctypes as C
        L.check(L.lib().evd_interpolate_poses(C.byref(self.struct), L.ptr(tt), n, L.ptr(p34), L.stream_ptr()), "evd_interpolate_poses")
        out[:, :3] = p34
        out[:, 3] = torch.tensor([0.0, 0.0, 0.0, 1.0], device=self.device)
        return out
