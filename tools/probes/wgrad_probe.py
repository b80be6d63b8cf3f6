#!/usr/bin/env python
"""Times the backward with the wgrad kernels ablated (tools/ablate.py variants of kernel_nerf_train_bwd_f16.hip): prints the
backward time on one stream; the difference between variants prices a part of k_wgrad.  GPU box only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["EVD_BWD_OVERLAP"] = "0"
import torch  # noqa: E402

from evdeblurnerf_amd import weights as W  # noqa: E402
from evdeblurnerf_amd.nerf import NeRF  # noqa: E402

net = NeRF(W.make_nerf_state_dict(21), precision="f16")
R, S = 4096, 128
rb = torch.randn((R, 11), device="cuda")
z = torch.linspace(0, 1, S, device="cuda").expand(R, S).contiguous()
d_raw = torch.randn((R, S, 4), device="cuda") * 1e-4
raw, store = net.mlpforward_train(rb, z)
for _ in range(2):
    net.mlp_backward_flat(d_raw, store)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(10):
    net.mlp_backward_flat(d_raw, store)
e1.record()
e1.synchronize()
print(f"backward (one stream): {e0.elapsed_time(e1) / 10:.3f} ms")
