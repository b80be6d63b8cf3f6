"""Timing of the compositing scan on 2^20 rays x 128 samples (GPU box): the form selected by the environment
(EVD_COMPOSITE_FORM=rows|stream, EVD_COMPOSITE_BPC, EVD_LIB_PATH for a variant library), its outputs' checksum, and torch's copy / sum
on the same bytes as the yardsticks.  tools/dev/composite_ab.sh runs the combinations."""
import os, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from evdeblurnerf_amd import _lib as L
Rc, S = 1 << 20, int(os.environ.get("PROBE_S", "128"))
torch.manual_seed(0)
raw = torch.randn((Rc, S, 4), device="cuda"); z = torch.linspace(0, 1, S, device="cuda").expand(Rc, S).contiguous(); rd = torch.randn((Rc, 3), device="cuda")
o3, o1, o2, ow = torch.empty((Rc, 3), device="cuda"), torch.empty(Rc, device="cuda"), torch.empty(Rc, device="cuda"), torch.empty((Rc, S), device="cuda")
def run(weights=True, act_rgb="sigmoid", sigma_ch=3):
    L.check(L.lib().evd_raw2outputs(L.ptr(raw), L.ptr(z), L.ptr(rd), 3, Rc, S, 4, sigma_ch, 0 if sigma_ch else 1, 3, L.ACT[act_rgb], L.ACT["relu"], 0, 0.0, None, L.ptr(o3), None, L.ptr(o2), L.ptr(ow) if weights else None, L.ptr(o1), None, 0, None, L.stream_ptr()))
def t(fn, n=20):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / n
rd_bytes = Rc * S * 20 + Rc * 12
tag = f"form={os.environ.get('EVD_COMPOSITE_FORM', 'default')} lib={os.path.basename(os.environ.get('EVD_LIB_PATH', 'default'))}"
for name, fn, b in (("full", lambda: run(), rd_bytes + Rc * S * 4 + Rc * 20), ("no weights store", lambda: run(False), rd_bytes + Rc * 20),
                    ("pdrf layout, rgb act none", lambda: run(True, "none", 0), rd_bytes + Rc * S * 4 + Rc * 20)):
    ms = t(fn); print(f"[{tag}] {name:26s} {ms:.3f} ms  {b / ms / 1e6:.0f} GB/s = {b / ms / 1e6 / 8000:.3f} of 8 TB/s")
run()
print(f"[{tag}] checksums rgb {o3.double().sum().item():.6f} depth {o1.double().sum().item():.6f} acc {o2.double().sum().item():.6f} weights {ow.double().sum().item():.6f}")
if os.environ.get("PROBE_YARDSTICKS"):
    x = torch.empty(Rc * S * 5, device="cuda"); y = torch.empty_like(x)
    ms = t(lambda: y.copy_(x)); print(f"torch copy 2.7GB    {ms:.3f} ms  {2 * x.numel() * 4 / ms / 1e6:.0f} GB/s (r+w)")
    ms = t(lambda: x.sum()); print(f"torch sum 2.7GB     {ms:.3f} ms  {x.numel() * 4 / ms / 1e6:.0f} GB/s (read)")
