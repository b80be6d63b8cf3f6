import os, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from evdeblurnerf_amd import _lib as L
Rc, S = 1 << 20, 128
raw = torch.randn((Rc, S, 4), device="cuda"); z = torch.linspace(0, 1, S, device="cuda").expand(Rc, S).contiguous(); rd = torch.randn((Rc, 3), device="cuda")
o3, o1, o2, ow = torch.empty((Rc, 3), device="cuda"), torch.empty(Rc, device="cuda"), torch.empty(Rc, device="cuda"), torch.empty((Rc, S), device="cuda")
def run(weights=True, act_rgb="sigmoid"):
    L.check(L.lib().evd_raw2outputs(L.ptr(raw), L.ptr(z), L.ptr(rd), 3, Rc, S, 4, 3, 0, 3, L.ACT[act_rgb], L.ACT["relu"], 0, 0.0, None, L.ptr(o3), None, L.ptr(o2), L.ptr(ow) if weights else None, L.ptr(o1), None, 0, None, L.stream_ptr()))
def t(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / n
rd_bytes = Rc * S * 20 + Rc * 12
for name, fn, b in (("full", lambda: run(), rd_bytes + Rc * S * 4 + Rc * 20), ("no weights store", lambda: run(False), rd_bytes + Rc * 20),
                    ("rgb act none", lambda: run(True, "none"), rd_bytes + Rc * S * 4 + Rc * 20)):
    ms = t(fn); print(f"{name:18s} {ms:.3f} ms  {b / ms / 1e6:.0f} GB/s")
x = torch.empty(Rc * S * 5, device="cuda"); y = torch.empty_like(x)
ms = t(lambda: y.copy_(x)); print(f"torch copy 2.7GB    {ms:.3f} ms  {2 * x.numel() * 4 / ms / 1e6:.0f} GB/s (r+w)")
ms = t(lambda: x.sum()); print(f"torch sum 2.7GB     {ms:.3f} ms  {x.numel() * 4 / ms / 1e6:.0f} GB/s (read)")
