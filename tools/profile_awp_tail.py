"""Steady-state kernel list of the AWP per-ray remainder (FusedAWP._per_ray with the MAM's per-sample part already reduced): forward +
backward at the blurfactory iteration's size, timed and broken down with torch.profiler.  GPU box only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from awp_standin import RefLikeAWP  # noqa: E402
from evdeblurnerf_amd.awp import FusedAWP  # noqa: E402

R, P, S = 1024, 10, 128
torch.manual_seed(0)
ref = RefLikeAWP(P=P, mam="corr").cuda().train()
fused = FusedAWP(ref, "f16")
h = torch.randn((R, P, 64), device="cuda", requires_grad=True)
view = torch.randn((R, ref.motion_feature_embed_layer[0].in_features - 64), device="cuda", requires_grad=True)
hi = torch.randn((R, P, 64), device="cuda", requires_grad=True)
hs = torch.randn((R, S, 64), device="cuda", requires_grad=True)
params = [p for p in ref.parameters() if p.requires_grad]


def step():
    out = fused._per_ray(h, view, None, R, P, S, hi, hs)
    torch.autograd.grad((out * out).sum(), [h, view, hi, hs] + params, allow_unused=True)


for _ in range(5):
    step()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(20):
    step()
e1.record()
e1.synchronize()
print(f"per-ray remainder forward + backward: {e0.elapsed_time(e1) / 20:.3f} ms")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(5):
        step()
    torch.cuda.synchronize()
ka = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)
tot = sum(e.device_time_total for e in ka)
print(f"device time per step {tot / 5 / 1e3:.3f} ms in {sum(e.count for e in ka) / 5:.0f} kernels")
for e in ka[:40]:
    print(f"{e.device_time_total / 5:9.1f} us  n={e.count / 5:5.1f}  {e.key[:120]}")
