"""The MotionAggregationModule at the blurfactory iteration's size (R = 1024 rays x P = 10 sub-exposures x S = 128 samples, h_local
[R P, S, 64]): forward + backward of (a) the plain torch module (tools/awp_standin.py MAMLike, the reference's structure: Linear over every
sample, Conv2d logit, two softmaxes, two weighted sums) and (b) FusedAWP._mam (per-sample part on evd_mam_local_forward / _backward), and
the two kernels alone with their HBM rates.  GPU box only."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from awp_standin import RefLikeAWP  # noqa: E402
from evdeblurnerf_amd.awp import FusedAWP, mam_local  # noqa: E402


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n


def run(R=1024, P=10, S=128):
    """-> dict: forward + backward of the plain torch module and of FusedAWP._mam, the two kernels alone"""
    torch.manual_seed(0)
    ref = RefLikeAWP(P=P, mam="corr").cuda()
    fused = FusedAWP(ref, "f16")
    h_local = torch.relu(torch.randn((R * P, S, 64), device="cuda")).requires_grad_(True)
    xg = torch.randn((R, P, 32), device="cuda", requires_grad=True)
    proj = torch.randn((R, P, 32), device="cuda")

    def both(f):
        out = f(xg, h_local)
        torch.autograd.grad((out * proj).sum(), [xg, h_local] + list(ref.MAM.parameters()), allow_unused=True)

    t_torch = timeit(lambda: both(ref.MAM))
    t_fused = timeit(lambda: both(lambda x, h: fused._mam(x, h, R, P, S)))
    mam = ref.MAM
    hd = h_local.detach()
    Wl, v = mam.linear.weight.detach(), mam.Corr.line_conv_att.weight.detach()
    t_f = timeit(lambda: mam_local(hd, Wl, v, R, P, S))
    hl = h_local.detach().requires_grad_(True)
    hi, hs = mam_local(hl, Wl, v, R, P, S)
    gi, gs = torch.randn_like(hi), torch.randn_like(hs)
    t_b = timeit(lambda: torch.autograd.grad([hi, hs], [hl], [gi, gs], retain_graph=True))
    nb = R * P * S * 64 * 4
    return {"R": R, "P": P, "S": S, "h_local_bytes": nb, "torch_fwd_bwd_ms": t_torch, "fused_fwd_bwd_ms": t_fused, "k_mam_local_fwd_ms": t_f,
            "k_mam_local_bwd_ms": t_b, "fwd_GBps": 2 * nb / t_f / 1e6, "bwd_GBps": 2 * nb / t_b / 1e6}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=1024)
    ap.add_argument("--P", type=int, default=10)
    ap.add_argument("--S", type=int, default=128)
    a = ap.parse_args()
    r = run(a.rays, a.P, a.S)
    print(f"MAM fwd+bwd at R={r['R']} P={r['P']} S={r['S']} (h_local {r['h_local_bytes'] / 2**20:.0f} MiB): torch {r['torch_fwd_bwd_ms']:.3f} ms | "
          f"fused {r['fused_fwd_bwd_ms']:.3f} ms ({r['torch_fwd_bwd_ms'] / r['fused_fwd_bwd_ms']:.1f}x); kernels alone: forward {r['k_mam_local_fwd_ms']:.3f} ms "
          f"({r['fwd_GBps']:.0f} GB/s over 2 reads of h_local), backward {r['k_mam_local_bwd_ms']:.3f} ms ({r['bwd_GBps']:.0f} GB/s over 1 read + 1 write)")


if __name__ == "__main__":
    main()
