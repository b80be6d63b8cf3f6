#!/usr/bin/env python
"""The x-y plane's scatter through the 8 x 8-cell window (k_scatter_xy, opt-in with EVD_SCATTER_WIN=1) against the all-atomics sweep: agreement and timing at the
blurfactory fine-level size, on rays along z (NDC-like: a tile's taps share a few cells) and on oblique rays (tiles overflow the window and
take the direct sweep).  GPU box only.   python tools/check_scatter_win.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from evdeblurnerf_amd import _lib as L, weights as W  # noqa: E402
from evdeblurnerf_amd.voxnerf import VoxelNeRFSampleFeatures, _grid_grads  # noqa: E402

AABB = ([-1.5, -1.5, -1.0], [1.5, 1.5, 1.0])


def main():
    voxels = 134217984
    g = W.pdrf_grid_size(AABB[0], AABB[1], voxels)
    sd = W.make_pdrf_state_dict(32, g, input_ch=127, hidden_dim=256, geo_feat_dim=128)
    net = VoxelNeRFSampleFeatures(sd, "", AABB, num_layers=2, hidden_dim=256, geo_feat_dim=128, num_layers_color=3, input_ch=127,
                                  app_dim=32, app_n_comp=(64, 16, 16), n_voxels=voxels)
    R, S = 4096, 128
    rs = np.random.RandomState(0)
    grids = net.grid_params()
    grads, gs = _grid_grads(net, grids)
    for name, spread in (("rays along z (NDC-like)", 0.02), ("oblique rays", 0.6), ("some samples outside the box", 0.02)):
        o = rs.uniform(-1.2, 1.2, (R, 1, 3)) * np.array([1, 1, 0]) + np.array([0, 0, 0.95])
        d = rs.normal(size=(R, 1, 3)) * spread + np.array([0, 0, -1.0])
        z = np.sort(rs.uniform(0.0, 1.9 if "outside" not in name else 2.3, (R, S, 1)), 1)
        pts = torch.as_tensor((o + d * z).astype(np.float32), device="cuda").reshape(-1, 3).contiguous()
        n = pts.shape[0]
        d_out = torch.randn((n, 32), device="cuda")
        d_pts = torch.empty((n, 3), device="cuda")
        nb = int(L.lib().evd_voxel_sample_bwd_workspace_bytes(net._h, n))
        ws = torch.empty((nb,), dtype=torch.uint8, device="cuda")
        direct = lambda: L.check(L.lib().evd_voxel_sample_bwd(net._h, L.ptr(pts), n, L.ptr(d_out), 32, 0, C.byref(gs), L.ptr(d_pts), L.stream_ptr()), "bwd")
        hybrid = lambda: L.check(L.lib().evd_voxel_sample_bwd_ws(net._h, L.ptr(pts), n, L.ptr(d_out), 32, 0, C.byref(gs), L.ptr(d_pts), L.ptr(ws), nb, L.stream_ptr()), "bwd_ws")

        def run(fn, win):
            os.environ["EVD_SCATTER_WIN"] = win
            for t in grads:
                t.zero_()
            fn()
            out = [t.clone() for t in grads] + [d_pts.clone()]
            for _ in range(2):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            e1.synchronize()
            return out, e0.elapsed_time(e1) / 10

        ref, t_ref = run(direct, "0")
        print(f"{name}: n = {n}; all taps by atomics {t_ref:.3f} ms")
        for label, fn, win in (("hybrid", hybrid, "0"), ("hybrid + x-y window", hybrid, "1")):
            out, t = run(fn, win)
            err = max(((a - b).norm() / b.norm().clamp_min(1e-30)).item() for a, b in zip(out, ref))
            print(f"    {label:20s} {t:.3f} ms   max relative L2 difference over the 7 grid gradients + d pts: {err:.1e}")
            assert err < 1e-5, err


if __name__ == "__main__":
    main()
