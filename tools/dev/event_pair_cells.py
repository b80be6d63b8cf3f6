#!/usr/bin/env python
"""VERDICT r5 item 3(b), by numbers: if an event's START and END ray went to one wavefront of the tri-plane scatter, how many of their plane taps
could be merged into one atomic request?  CPU, numpy.  The two rays of an event see the same pixel from two camera poses a few milliseconds apart
(reference data/loader_events.py:280-306; here: a pose step of the size one exposure's blur kernel spans, divided by the events per exposure), and are
sampled as the reference samples every training ray: stratified depths with perturb = 1 (run_nerf.py: perturb default 1.0; renderer.py:163-178), i.e.
an independent uniform draw per ray and bin.  A tap of the x-z / y-z planes (16 channels = one 64-byte atomic request each, 80 % of the scatter's
requests) can merge only if both rays' k-th samples fall into the same (x or y, z) cell; a tap of the x-y plane if they share the (x, y) cell.
    python tools/dev/event_pair_cells.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from evdeblurnerf_amd import weights as W

AABB = np.array(W.BLURFACTORY_AABB, np.float64)
GRID = np.array(W.pdrf_grid_size(W.BLURFACTORY_AABB[0], W.BLURFACTORY_AABB[1], W.BLURFACTORY_FINE_VOXELS), np.float64)


def ndc_points(o, d, z):
    """NDC rays (utils/rays.py ndc_rays restated for H = W = 400, focal = 400, near = 1) -> points [R, S, 3]"""
    H = Wd = 400.0; f = 400.0; near = 1.0
    t = -(near + o[:, 2]) / d[:, 2]
    o = o + t[:, None] * d
    o0 = -1.0 / (Wd / (2.0 * f)) * o[:, 0] / o[:, 2]; o1 = -1.0 / (H / (2.0 * f)) * o[:, 1] / o[:, 2]; o2 = 1.0 + 2.0 * near / o[:, 2]
    d0 = -1.0 / (Wd / (2.0 * f)) * (d[:, 0] / d[:, 2] - o[:, 0] / o[:, 2]); d1 = -1.0 / (H / (2.0 * f)) * (d[:, 1] / d[:, 2] - o[:, 1] / o[:, 2]); d2 = -2.0 * near / o[:, 2]
    on = np.stack([o0, o1, o2], -1); dn = np.stack([d0, d1, d2], -1)
    return on[:, None, :] + dn[:, None, :] * z[..., None]


def cells(p):
    u = (p - AABB[0]) / (AABB[1] - AABB[0]) * (GRID - 1.0)            # align_corners=True grid coordinates
    return np.floor(u).astype(np.int64)


def main():
    rs = np.random.RandomState(0)
    R, S = 4096, 128
    rays = W.synthetic_rays(3, R).astype(np.float64)                   # [R, 3, 2]
    o, d = rays[..., 0], rays[..., 1]
    for step_deg, step_t, label in ((0.02, 0.0005, "one event interval (exposure motion / ~50 events per pixel)"), (0.2, 0.005, "a tenth of the exposure's motion"),
                                    (0.0, 0.0, "identical poses (upper bound)")):
        # the end pose: a small rotation about y + translation along x of the start pose
        a = np.deg2rad(step_deg)
        Rm = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        o2, d2 = o + np.array([step_t, 0, 0]), d @ Rm.T
        for perturb in (1, 0):
            edges = np.linspace(0.0, 1.0, S + 1)
            lo, hi = edges[:-1], edges[1:]
            def z(r):
                return (lo + (hi - lo) * (r.uniform(size=(R, S)) if perturb else 0.5))
            c1, c2 = cells(ndc_points(o, d, z(rs))), cells(ndc_points(o2, d2, z(rs)))
            same = lambda ax: np.all(c1[..., ax] == c2[..., ax], -1).mean()
            xy, xz, yz = same([0, 1]), same([0, 2]), same([1, 2])
            # requests per sample pair without merging: x-y plane 4 taps x 4 segments of 64 bytes (run-merged along the ray ~ / 10), x-z and y-z 4 taps x 1 segment each
            base = 2 * (16 / 10.0 + 4 + 4)
            merged = base - (xy * 16 / 10.0 + xz * 4 + yz * 4)
            print(f"{label:62s} perturb={perturb}: same cell of the pair's k-th samples: x-y {100 * xy:5.1f} %  x-z {100 * xz:5.1f} %  y-z {100 * yz:5.1f} %"
                  f"  -> atomic requests of the pair {base:.1f} -> {merged:.1f} ({100 * (1 - merged / base):.0f} % fewer on the event rays, {44 * (1 - merged / base):.0f} % of the iteration's)")
    print("(the fine level's importance samples are drawn per ray from its own coarse weights: they differ between the two rays even with perturb = 0)")


if __name__ == "__main__":
    main()
