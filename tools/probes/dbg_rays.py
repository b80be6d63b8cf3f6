import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from evdeblurnerf_amd import weights as W
from evdeblurnerf_amd.nerf import NeRF
from test_gpu_train import make_inputs, rel_l2
from torch_restatement import TorchNerf
sd = W.make_nerf_state_dict(23)
R, S = 48, 40
rb_np, z_np = make_inputs(R, S, 12)
wgt = (np.random.RandomState(13).normal(size=(R, S, 4)) * 1e-2).astype(np.float32)
pts64 = torch.tensor(rb_np[:, None, 0:3].astype(np.float64) + rb_np[:, None, 3:6].astype(np.float64) * z_np[..., None].astype(np.float64)).reshape(-1, 3).requires_grad_(True)
dirs = torch.tensor(np.repeat(rb_np[:, None, 8:11], S, 1), dtype=torch.float64).reshape(-1, 3)
ref = TorchNerf(sd)
(ref(pts64, dirs) * torch.tensor(wgt, dtype=torch.float64).reshape(-1, 4)).sum().backward()
for prec in ("f16", "f16x3"):
    net = NeRF(sd, precision=prec)
    rb = torch.tensor(rb_np, device="cuda"); z = torch.tensor(z_np, device="cuda")
    raw, store = net.mlpforward_train(rb, z)
    pts = (rb[:, None, 0:3] + rb[:, None, 3:6] * z[..., None]).reshape(-1, 3).contiguous()
    out = net.mlp_backward_flat(torch.tensor(wgt, device="cuda"), store, prec, pts=pts, ray_batch=rb)
    g, d_pts, d_dirs = out
    e = (d_pts.cpu().double() - pts64.grad)
    print(prec, "d_pts rel", rel_l2(d_pts.cpu().double(), pts64.grad), "flat nan", torch.isnan(g).sum().item(), "worst rows", e.abs().max(1)[0].topk(5))
    print("  per component rel:", [rel_l2(d_pts.cpu().double()[:, c], pts64.grad[:, c]) for c in range(3)])
    print("  per tile rel (first 8):", [round(rel_l2(d_pts.cpu().double()[t*32:(t+1)*32], pts64.grad[t*32:(t+1)*32]), 6) for t in range(8)])
