#!/usr/bin/env python
"""Timing of the tri-plane gather and its scatter-add backward at the blurfactory fine-level size. GPU box only.
    python tools/bench_voxel_bwd.py [--rays 4096] [--samples 128] [--voxels 134217984]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

from evdeblurnerf_amd import _lib as L, weights as W  # noqa: E402
from evdeblurnerf_amd.voxnerf import VoxelNeRFSampleFeatures, _grid_grads  # noqa: E402

AABB = ([-1.5, -1.5, -1.0], [1.5, 1.5, 1.0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--voxels", type=int, default=134217984)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--slope", type=float, default=0.35, help="std of the rays' x / y direction components (0: rays along z, as an NDC scene's reference camera)")
    ap.add_argument("--dpts", action="store_true", help="also the gradient w.r.t. the sample positions (the blur batch)")
    a = ap.parse_args()
    g = W.pdrf_grid_size(AABB[0], AABB[1], a.voxels)
    sd = W.make_pdrf_state_dict(32, g, input_ch=127, hidden_dim=256, geo_feat_dim=128)
    net = VoxelNeRFSampleFeatures(sd, "", AABB, num_layers=2, hidden_dim=256, geo_feat_dim=128, num_layers_color=3, input_ch=127,
                                  app_dim=32, app_n_comp=(64, 16, 16), n_voxels=a.voxels)
    R, S = a.rays, a.samples
    rs = np.random.RandomState(0)
    o = rs.uniform(-0.3, 0.3, (R, 1, 3)) + np.array([0, 0, 0.9])
    d = rs.normal(size=(R, 1, 3)) * a.slope * np.array([1.0, 1.0, 0.0]) + np.array([0, 0, -1.0])
    z = np.sort(rs.uniform(0.1, 1.7, (R, S, 1)), 1)
    pts = torch.as_tensor((o + d * z).astype(np.float32), device="cuda").reshape(-1, 3).contiguous()
    n = pts.shape[0]
    d_out = torch.randn((n, 32), device="cuda")
    grids = net.grid_params()
    grads, gs = _grid_grads(net, grids)
    dp = torch.empty_like(pts) if a.dpts else None

    def timed(fn):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / a.iters

    fwd = timed(lambda: net.sample(pts))
    bwd = timed(lambda: L.check(L.lib().evd_voxel_sample_bwd(net._h, L.ptr(pts), n, L.ptr(d_out), 32, 0, C.byref(gs), L.ptr(dp), L.stream_ptr()), "bwd"))
    atom = n * 6 * 96
    nb = int(L.lib().evd_voxel_sample_bwd_workspace_bytes(net._h, n))
    ws = torch.empty((nb,), dtype=torch.uint8, device="cuda")
    binned = timed(lambda: L.check(L.lib().evd_voxel_sample_bwd_ws(net._h, L.ptr(pts), n, L.ptr(d_out), 32, 0, C.byref(gs), L.ptr(dp), L.ptr(ws), nb, L.stream_ptr()), "bwd_ws"))
    # agreement of the two forms (both sum in a non-deterministic order)
    for t in grads:
        t.zero_()
    L.check(L.lib().evd_voxel_sample_bwd(net._h, L.ptr(pts), n, L.ptr(d_out), 32, 0, C.byref(gs), L.ptr(dp), L.stream_ptr()), "bwd")
    ref = [t.clone() for t in grads]
    for t in grads:
        t.zero_()
    L.check(L.lib().evd_voxel_sample_bwd_ws(net._h, L.ptr(pts), n, L.ptr(d_out), 32, 0, C.byref(gs), L.ptr(dp), L.ptr(ws), nb, L.stream_ptr()), "bwd_ws")
    err = max(((a - b).norm() / b.norm().clamp_min(1e-30)).item() for a, b in zip(grads, ref))
    print(f"[slope {a.slope}, d pts {a.dpts}, EVD_SCATTER_FORM={os.environ.get('EVD_SCATTER_FORM', 'wave')}] grid {g}: n = {n} samples | gather forward {fwd:.3f} ms | scatter backward: direct atomics {bwd:.3f} ms = {atom / bwd / 1e6:.1f} G float atomics/s | "
          f"with scratch (hybrid: lines through LDS slices; EVD_SCATTER=binned: sort + LDS tiles) {binned:.3f} ms | relative L2 difference of the two {err:.1e} | scratch {nb / 2**20:.0f} MiB")


if __name__ == "__main__":
    main()
