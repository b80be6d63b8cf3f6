"""The AWP's per-ray remainder (evd_awp_tail_forward / _backward: awp.py:89-95, 104-117, mam.py:35-53) at the blurfactory shape -- 1024 rays,
P = 10 sub-exposures, 64 + 64 samples, view embedding 32 + 15 direction columns -- on the kernels and on round 3's channel-last torch path,
with the float64-autograd error of both.  GPU box only.    python tools/bench_awp_tail.py"""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def _time(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def run(R=1024, P=10, S=128, VF=32):
    from awp_standin import RefLikeAWP
    from evdeblurnerf_amd.awp import FusedAWP
    torch.manual_seed(1)
    base = RefLikeAWP(P=P, view_ch=VF, mam="corr").cuda().train()
    with torch.no_grad():                                      # attention logits of order 1
        for conv, k in ((base.MAM.Corr.conva, 6.0), (base.MAM.Corr.convb, 6.0), (base.MAM.Corr.convc, 6.0)):
            conv.weight.mul_(k)
    rs = np.random.RandomState(2)
    t = lambda a: torch.tensor(a.astype(np.float32)).cuda()
    h, hi, hs = t(rs.standard_normal((R, P, 64))), t(np.abs(rs.standard_normal((R, P, 64)))), t(np.abs(rs.standard_normal((R, S, 64))))
    vf, rd, proj = t(rs.standard_normal((R, VF))), t(rs.standard_normal((R * P, 3))), t(rs.standard_normal((R, P)))
    out = {}
    res = {}
    for tag, kern in (("kernels", True), ("torch", False)):
        m = copy.deepcopy(base)
        fused = FusedAWP(m, tail_kernels=kern)
        ins = [x.clone().requires_grad_(True) for x in (h, rd, hi, hs, vf)]
        ps = fused._tail_params()

        def fwd():
            if kern:
                return fused._tail(ins[0], ins[4], ins[1], ins[2], ins[3], R, P, S)
            dirs = ins[1].reshape(R, P, -1)[:, 0, :]
            view = torch.cat([ins[4], m.ray_dirs_embed_fn(dirs / torch.norm(dirs, dim=-1, keepdim=True))], -1)
            return fused._per_ray(ins[0], view, None, R, P, S, ins[2], ins[3])

        def both():
            o = fwd()
            torch.autograd.grad((o * proj).sum(), ins + ps)

        with torch.no_grad():
            t_f = _time(fwd)
        t_fb = _time(both)
        o = fwd()
        res[tag] = (o.detach(), torch.autograd.grad((o * proj).sum(), ins + ps))
        out[tag] = {"forward_us": t_f, "forward_backward_us": t_fb}
    m64 = copy.deepcopy(base).double()
    f64 = FusedAWP.__new__(FusedAWP)
    torch.nn.Module.__init__(f64)
    f64.ref = m64
    ins64 = [x.double().clone().requires_grad_(True) for x in (h, rd, hi, hs, vf)]
    dirs = ins64[1].reshape(R, P, -1)[:, 0, :]
    view = torch.cat([ins64[4], m64.ray_dirs_embed_fn(dirs / torch.norm(dirs, dim=-1, keepdim=True))], -1)
    o64 = FusedAWP._per_ray(f64, ins64[0], view, None, R, P, S, ins64[2], ins64[3])
    ps64 = FusedAWP._tail_params(f64)
    g64 = torch.autograd.grad((o64 * proj.double()).sum(), ins64 + ps64)
    for tag in res:
        o, g = res[tag]
        worst = max(float((a.double() - b).norm() / b.norm().clamp_min(1e-30)) for a, b in zip(g, g64) if float(b.norm()) > 1e-9 * float(g64[0].norm()))
        out[tag]["out_linf_vs_float64"] = float((o.double() - o64.detach()).abs().max())
        out[tag]["worst_gradient_error_of_norm_vs_float64_autograd"] = worst
    out["shape"] = {"rays": R, "P": P, "S": S, "view_feature": VF, "direction_freqs": 2}
    out["launches"] = {"kernels": "2 forward + 3 backward library launches, no torch launch", "torch": "~60 aten launches each way"}
    return out


if __name__ == "__main__":
    import json
    print(json.dumps(run(), indent=1))
