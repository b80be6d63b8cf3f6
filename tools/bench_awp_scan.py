"""The AWP consumer's streaming kernels at the blurfactory iteration's size (1024 rays x 10 sub-exposures x 128 samples x 64 channels,
h_local = 335 MB): feature_integration forward / backward (k_awp_integrate, k_awp_integrate_bwd) and the MAM's per-sample part
(k_mam_local_fwd / _bwd), with their HBM rates.  GPU box only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evdeblurnerf_amd.awp import feature_integration, mam_local  # noqa: E402


def timeit(fn, n=30, warm=5):
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


R, P, S, Cc = 1024, 10, 128, 64
torch.manual_seed(0)
h = torch.relu(torch.randn((R, P, S, Cc), device="cuda")) * 3.0
z = torch.sort(torch.rand((R * P, S), device="cuda"), -1).values
d = torch.randn((R * P, 3), device="cuda")
nb = h.numel() * 4
t_f = timeit(lambda: feature_integration(h, z, d))
hg = h.clone().requires_grad_(True)
out = feature_integration(hg, z, d)
g = torch.randn_like(out)
t_b = timeit(lambda: torch.autograd.grad(out, hg, g, retain_graph=True))
Wl, v = torch.randn((32, 64), device="cuda") * 0.1, torch.randn((1, 32, 1, 1), device="cuda") * 0.1
hl = h.reshape(R * P, S, Cc)
t_mf = timeit(lambda: mam_local(hl, Wl, v, R, P, S))
hlg = hl.clone().requires_grad_(True)
hi, hs = mam_local(hlg, Wl, v, R, P, S)
gi, gs = torch.randn_like(hi), torch.randn_like(hs)
t_mb = timeit(lambda: torch.autograd.grad([hi, hs], [hlg], [gi, gs], retain_graph=True))
print(f"h_local {nb / 2**20:.0f} MiB | integrate fwd {t_f * 1e3:.0f} us ({nb / t_f / 1e9:.2f} TB/s) | integrate bwd {t_b * 1e3:.0f} us ({2 * nb / t_b / 1e9:.2f} TB/s) | "
      f"mam_local fwd {t_mf * 1e3:.0f} us ({nb / t_mf / 1e9:.2f} TB/s) | mam_local bwd {t_mb * 1e3:.0f} us ({2 * nb / t_mb / 1e9:.2f} TB/s)")
