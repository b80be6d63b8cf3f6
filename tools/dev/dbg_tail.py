import sys, os, copy
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo")
import numpy as np, torch
from awp_standin import RefLikeAWP
from evdeblurnerf_amd.awp import FusedAWP
torch.manual_seed(5)
rs = np.random.RandomState(17)
R, P, S = 64, 5, 32
ref = RefLikeAWP(P=P, mam="corr").cuda()
ref2 = RefLikeAWP(P=P, mam="corr").cuda()
ref2.load_state_dict(ref.state_dict())
A, B = FusedAWP(ref, "f16"), FusedAWP(ref2, "f16", tail_kernels=False)
opts = [torch.optim.SGD(m.parameters(), lr=1e-2) for m in (ref, ref2)]
_t = lambda a: torch.tensor(a).cuda()
import copy
cap = {}
orig = A._tail
def spy(h, vf, rd, hi, hs, n_ray, P_, S_):
    cap.update(h=h.detach().clone(), vf=vf.detach().clone(), rd=rd.detach().clone(), hi=hi.detach().clone(), hs=hs.detach().clone())
    return orig(h, vf, rd, hi, hs, n_ray, P_, S_)
A._tail = spy
for step in range(3):
    df = _t((0.5 * rs.standard_normal((R * P, S, 128))).astype(np.float32))
    z = _t(np.sort(rs.uniform(0, 1, (R * P, S)).astype(np.float32), -1))
    d = _t(rs.standard_normal((R * P, 3)).astype(np.float32))
    vf = _t(rs.standard_normal((R, 4)).astype(np.float32))
    proj = _t(rs.standard_normal((R, P)).astype(np.float32))
    outs, grads = [], []
    for fused, m, opt in ((A, ref, opts[0]), (B, ref2, opts[1])):
        opt.zero_grad(set_to_none=True)
        df_ = df.clone().requires_grad_(True)
        dd = d.clone().requires_grad_(True)
        out = fused(df_, z, dd, vf)
        (out * proj).sum().backward()
        outs.append(out.detach().clone())
        grads.append({n_: (p.grad.detach().clone() if p.grad is not None else None) for n_, p in m.named_parameters()})
        grads[-1]["d depth_feature"] = df_.grad.detach().clone()
        grads[-1]["d rays_d"] = dd.grad.detach().clone()
        if fused is A:
            m64 = copy.deepcopy(ref).double()
            for p_ in m64.parameters(): p_.grad = None
            F64 = FusedAWP.__new__(FusedAWP); torch.nn.Module.__init__(F64); F64.ref = m64
            h, hi, hs = (cap[k].double().requires_grad_(True) for k in ("h", "hi", "hs"))
            dirs = cap["rd"].double().reshape(R, P, -1)[:, 0, :]
            view = torch.cat([cap["vf"].double(), m64.ray_dirs_embed_fn(dirs / torch.norm(dirs, dim=-1, keepdim=True))], -1)
            o64 = FusedAWP._per_ray(F64, h, view, None, R, P, S, hi, hs)
            (o64 * proj.double()).sum().backward()
            g64 = {n_: p_.grad.detach().clone() for n_, p_ in m64.named_parameters() if p_.grad is not None}
        opt.step()
    for n_ in ("MAM.Corr.convd.1.bias", "MAM.Corr.convd.1.weight", "MAM.Corr.convd.0.weight", "MAM.Corr.conva.weight", "motion_feature_embed_layer.0.weight"):
        r64 = g64[n_]
        print(f"   vs f64: {n_:40s} kernels {float((grads[0][n_].double() - r64).norm() / r64.norm()):.2e}   torch {float((grads[1][n_].double() - r64).norm() / r64.norm()):.2e}")
    ref2.load_state_dict(ref.state_dict())
    print("step", step, "out", float((outs[0] - outs[1]).abs().max()))
    for n_ in grads[0]:
        a, b = grads[0][n_], grads[1][n_]
        if a is None or b is None: continue
        print(f"   {n_:45s} {float((a - b).abs().max()) / (float(a.abs().max()) + 1e-12):.2e}  (max {float(a.abs().max()):.2e})")

