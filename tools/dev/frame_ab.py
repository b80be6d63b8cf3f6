"""Full 400 x 400 frame render (c2f, 64 + 128 samples, f16c): ms per frame and the gather kernels' share, for developer A/B runs of
library variants (EVD_LIB_PATH).  Round 5 used it for an XCD-aware block -> sample-group map of k_voxel_sample_w (XCD x takes the contiguous
chunk x of a raster-ordered frame's samples instead of every eighth group): 33.29 vs 33.29 ms per frame, gathers 11.61 vs 11.66 ms --
no effect (the grids sit in the Infinity Cache, the kernel is a latency chain), removed.  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from evdeblurnerf_amd import weights as W
from evdeblurnerf_amd.renderer import NeRFAll
model = NeRFAll(W.blurfactory_args(64), W.make_blurfactory_state_dict(31), precision="f16c").eval()
K = W.synthetic_camera()
poses = [torch.as_tensor(W.synthetic_pose(40 + i)[:3, :4].astype(np.float32), device="cuda") for i in range(4)]
kw = dict(ndc=True, near=0., far=1., use_viewdirs=True, N_samples=64, N_importance=128, perturb=0., raw_noise_std=0.)
model.render_path(400, 400, K, 1 << 22, poses[:1], kw)
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    rgbs, _ = model.render_path(400, 400, K, 1 << 22, poses, kw)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / len(poses))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    model.render_path(400, 400, K, 1 << 22, poses[:2], kw)
    torch.cuda.synchronize()
ka = [e for e in prof.key_averages() if e.device_time_total > 0]
g = sum(e.device_time_total for e in ka if "k_voxel_sample_w" in e.key) / 2e3
tot = sum(e.device_time_total for e in ka) / 2e3
print(f"lib={os.path.basename(os.environ.get('EVD_LIB_PATH', 'default'))}: {1e3 * best:.3f} ms per frame; device time {tot:.3f} ms per frame, gathers {g:.3f} ms; checksum {float(torch.as_tensor(rgbs).double().sum()):.6f}")
