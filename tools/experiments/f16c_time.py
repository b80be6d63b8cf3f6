import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from evdeblurnerf_amd import weights as W
from evdeblurnerf_amd.nerf import NeRF
from evdeblurnerf_amd.renderer import NeRFAll
dev = "cuda"; R, S = 4096, 128
rays = torch.as_tensor(W.synthetic_rays(100, R), device=dev)
rb = NeRFAll.ray_batch_train(400, 400, W.synthetic_camera(), rays).contiguous()
z = torch.linspace(0, 1, S, device=dev).expand(R, S).contiguous()
net = NeRF(W.make_nerf_state_dict(21))
for prec in sys.argv[1:] or ["f16c"]:
    for _ in range(5): net.mlpforward(rb, z, precision=prec)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): net.mlpforward(rb, z, precision=prec)
    e1.record(); torch.cuda.synchronize()
    print(f"DBG={os.environ.get('EVD_F16C_DBG', '0'):>4s} {prec:6s} {e0.elapsed_time(e1) / 30:.4f} ms")
