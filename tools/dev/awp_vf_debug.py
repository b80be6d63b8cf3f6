"""debug: d view_feature of the AWP tail kernels at G32's shape against float64 autograd"""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from test_gpu_awp_tail import _random_case, _kernel_chain, rel
from evdeblurnerf_amd.awp import FusedAWP
for (R, P, S, VF) in [(32, 5, 32, 32), (32, 5, 64, 32), (300, 10, 128, 32), (32, 5, 32, 4)]:
    awp, h_local, z, rays_d, vf, proj = _random_case(R, P, S, VF, seed=5)
    ref = copy.deepcopy(awp).double().train()
    awp = awp.cuda().train()
    fused = FusedAWP(awp)
    dev = lambda a: torch.tensor(a, dtype=torch.float32).cuda().requires_grad_(True)
    hl, rd, v = dev(h_local), dev(rays_d), dev(vf)
    out = _kernel_chain(fused, hl, torch.tensor(z).cuda(), rd, v, R, P, S)
    g = torch.autograd.grad((out * torch.tensor(proj).cuda()).sum(), [hl, rd, v])
    c64 = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
    hl64, rd64, v64 = c64(h_local), c64(rays_d), c64(vf)
    o64 = ref.forward_from_local(hl64, torch.tensor(z, dtype=torch.float64), rd64, v64)
    g64 = torch.autograd.grad((o64 * torch.tensor(proj, dtype=torch.float64)).sum(), [hl64, rd64, v64])
    print((R, P, S, VF), "out", (out.detach().cpu().double() - o64.detach()).abs().max().item(), "grads", [rel(a, b) for a, b in zip(g, g64)])
    # through FusedAWP.forward on float rows
    x = torch.randn((R * P, S, 128), device="cuda", requires_grad=True)
    v2 = dev(vf)
    o2 = fused(x, torch.tensor(z).cuda(), torch.tensor(rays_d).cuda(), v2)
    (o2 * torch.tensor(proj).cuda()).sum().backward()
    x3 = x.detach().clone().requires_grad_(True); v3 = dev(vf)
    o3 = awp(x3, torch.tensor(z).cuda(), torch.tensor(rays_d).cuda(), v3)
    (o3 * torch.tensor(proj).cuda()).sum().backward()
    print("   module forward: d view_feature fused vs torch", rel(v2.grad, v3.grad), "d x", rel(x.grad, x3.grad))
