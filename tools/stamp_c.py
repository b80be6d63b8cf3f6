"""Developer probe: where does a wavefront of the compensated-float16 NeRF kernel wait?  Needs a library built with -DEVD_C_STAMP
(mlp_pipe_c.h: chunk_end accumulates the shader-clock cycles of its vmcnt wait and of its barrier; lane 0 of every wavefront writes
them in place of its sample).  EVD_LIB_PATH=.../libevdnerf_stamp.so python tools/stamp_c.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from evdeblurnerf_amd import weights as W
from evdeblurnerf_amd.nerf import NeRF

R, S = 4096, 128
net = NeRF(W.make_nerf_state_dict(21))
rs = np.random.RandomState(0)
rb = np.zeros((R, 11), np.float32)
rb[:, :3] = rs.uniform(-1, 1, (R, 3)); rb[:, 3:6] = rs.uniform(-1, 1, (R, 3)); rb[:, 7] = 1
vd = rs.standard_normal((R, 3)); rb[:, 8:] = vd / np.linalg.norm(vd, axis=-1, keepdims=True)
rb = torch.as_tensor(rb, device="cuda")
z = torch.linspace(0, 1, S, device="cuda").expand(R, S).contiguous()
for _ in range(3):
    raw, _ = net.mlpforward(rb, z, precision="f16c")
torch.cuda.synchronize()
r = raw.reshape(-1, 4)[::32].cpu().numpy()          # lane 0 of every wavefront
assert (r[:, 3] == -1).all(), "library was not built with -DEVD_C_STAMP"
tw, tb, tot = r[:, 0], r[:, 1], r[:, 2]
print(f"wavefronts {len(r)}: kernel cycles per wavefront mean {tot.mean():.0f} (min {tot.min():.0f}, max {tot.max():.0f})")
print(f"  vmcnt wait at chunk ends: mean {tw.mean():.0f} cycles = {100 * tw.mean() / tot.mean():.1f} %   (p10 {np.percentile(tw, 10):.0f}, p90 {np.percentile(tw, 90):.0f})")
print(f"  barrier at chunk ends:    mean {tb.mean():.0f} cycles = {100 * tb.mean() / tot.mean():.1f} %   (p10 {np.percentile(tb, 10):.0f}, p90 {np.percentile(tb, 90):.0f})")
