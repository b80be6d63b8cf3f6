"""Developer check: the AWP tail kernels on odd shapes (one ray, one sub-exposure, one sample, widths at the limits) against float64."""
import sys, os, copy
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo")
import numpy as np, torch
import test_gpu_awp_tail as T
from evdeblurnerf_amd.awp import FusedAWP, feature_integration, mam_local
for (R, P, S, VF) in ((1, 2, 1, 0), (1, 2, 2, 1), (2, 3, 5, 3), (3, 16, 96, 64), (5, 10, 130, 32), (4, 10, 160, 32), (257, 4, 16, 7), (1, 10, 128, 32)):
    awp, h_local, z, rays_d, vf, proj = T._random_case(R, P, S, VF, seed=R * 7 + S)
    ref = copy.deepcopy(awp).double().train()
    awp = awp.cuda().train()
    fused = FusedAWP(awp)
    dev = lambda a: None if a is None else torch.tensor(a).cuda().requires_grad_(True)
    hl, rd, v = dev(h_local), dev(rays_d), dev(vf)
    mam = awp.MAM
    h = feature_integration(hl.reshape(R, P, S, -1), torch.tensor(z).cuda(), rd)
    hi, hs = mam_local(hl, mam.linear.weight, mam.Corr.line_conv_att.weight, R, P, S)
    out = fused._tail(h, v, rd, hi, hs, R, P, S)
    if out is None:
        print(f"R {R} P {P} S {S} VF {VF}: refused by the library (LDS) -> torch remainder")
        continue
    ins = [hl, rd] + ([v] if v is not None else [])
    pk = [p_ for n_, p_ in awp.named_parameters() if not n_.startswith(("sample_feature_embed_layer", "MAM.conv."))]
    grads = torch.autograd.grad((out * torch.tensor(proj).cuda()).sum(), ins + pk)
    c64 = lambda a: None if a is None else torch.tensor(a, dtype=torch.float64, requires_grad=True)
    hl64, rd64, v64 = c64(h_local), c64(rays_d), c64(vf)
    out64 = ref.forward_from_local(hl64, torch.tensor(z, dtype=torch.float64), rd64, v64)
    names = [n_ for n_, _ in ref.named_parameters() if not n_.startswith(("sample_feature_embed_layer", "MAM.conv."))]
    pr = dict(ref.named_parameters())
    g64 = torch.autograd.grad((out64 * torch.tensor(proj, dtype=torch.float64)).sum(), [hl64, rd64] + ([v64] if v64 is not None else []) + [pr[n_] for n_ in names], allow_unused=True)
    scale = max(float(g_.norm()) for g_ in g64 if g_ is not None)
    worst = 0.0
    for a, b in zip(grads, g64):
        if b is None or float(b.norm()) < 1e-6 * scale:
            continue
        worst = max(worst, T.rel(a, b))
    print(f"R {R} P {P} S {S} VF {VF}: out {float((out.detach().cpu().double() - out64.detach()).abs().max()):.1e}, worst gradient {worst:.1e}, finite {bool(torch.isfinite(out).all())}")
