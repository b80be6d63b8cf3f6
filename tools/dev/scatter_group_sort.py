"""Developer experiment: does the tri-plane scatter get cheaper when the samples of the P sub-exposure rays of a pixel (nearly the same line
through the volume, different z per ray) are handed over SORTED BY z across the group -- so that the kernel's run-length merging sees runs
across rays?  Physical permutation of the points / gradient rows here; GPU box only."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from evdeblurnerf_amd import _lib as L, weights as W
from evdeblurnerf_amd.voxnerf import VoxelNeRFSampleFeatures, _grid_grads

AABB = ([-1.5, -1.5, -1.0], [1.5, 1.5, 1.0])
voxels = 134217984
g = W.pdrf_grid_size(AABB[0], AABB[1], voxels)
sd = W.make_pdrf_state_dict(32, g, input_ch=127, hidden_dim=256, geo_feat_dim=128)
net = VoxelNeRFSampleFeatures(sd, "", AABB, num_layers=2, hidden_dim=256, geo_feat_dim=128, num_layers_color=3, input_ch=127, app_dim=32, app_n_comp=(64, 16, 16), n_voxels=voxels)
grads, gs = _grid_grads(net, net.grid_params())
rs = np.random.RandomState(0)


def timed(fn, iters=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


for (R, P, S, jitter) in ((1024, 10, 128, 0.002), (1024, 10, 128, 0.01), (1024, 10, 64, 0.002), (4096, 1, 128, 0.0)):
    o = rs.uniform(-0.3, 0.3, (R, 1, 1, 3)) + np.array([0, 0, 0.9])
    d = rs.normal(size=(R, 1, 1, 3)) * 0.05 * np.array([1.0, 1.0, 0.0]) + np.array([0, 0, -1.0])
    o = o + rs.normal(size=(R, P, 1, 3)) * jitter * np.array([1.0, 1.0, 0.0])          # the sub-exposures: the same ray, moved by the blur kernel
    d = d + rs.normal(size=(R, P, 1, 3)) * jitter * np.array([1.0, 1.0, 0.0])
    z = np.sort(rs.uniform(0.1, 1.7, (R, P, S, 1)), 2)                                   # every ray its own stratified z
    pts_np = (o + d * z).astype(np.float32)                                             # [R, P, S, 3]
    pts = torch.as_tensor(pts_np, device="cuda").reshape(-1, 3).contiguous()
    n = pts.shape[0]
    d_out = torch.randn((n, 32), device="cuda")
    order = torch.as_tensor(np.argsort(z.reshape(R, P * S), axis=1), device="cuda")      # within a pixel's P S samples, by z
    perm = (order + torch.arange(R, device="cuda")[:, None] * (P * S)).reshape(-1)
    nb = int(L.lib().evd_voxel_sample_bwd_workspace_bytes(net._h, n))
    ws = torch.empty((nb,), dtype=torch.uint8, device="cuda")
    res = {}
    for tag, (pp, gg_) in {"ray order": (pts, d_out), "sorted over the pixel's rays": (pts[perm].contiguous(), d_out[perm].contiguous())}.items():
        for dp in (None, torch.empty_like(pp)):
            t = timed(lambda: L.check(L.lib().evd_voxel_sample_bwd_ws(net._h, L.ptr(pp), n, L.ptr(gg_), 32, 0, C.byref(gs), L.ptr(dp), L.ptr(ws), nb, L.stream_ptr()), "bwd_ws"))
            res[(tag, dp is not None)] = t
    t_perm = timed(lambda: (pts[perm].contiguous(), d_out[perm].contiguous()))
    print(f"R {R} P {P} S {S} jitter {jitter}: n = {n} | ray order {res[('ray order', False)]:.3f} / {res[('ray order', True)]:.3f} ms (grids only / + d pts) | "
          f"sorted {res[('sorted over the pixel' + chr(39) + 's rays', False)]:.3f} / {res[('sorted over the pixel' + chr(39) + 's rays', True)]:.3f} ms | the two physical gathers {t_perm:.3f} ms")
