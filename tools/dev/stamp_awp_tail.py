"""Developer probe: where do the AWP per-ray kernels (k_awp_tail_fwd / k_awp_tail_bwd) spend a ray?  Needs kernel_awp_tail.hip built with
-DEVD_AT_STAMP (tools/dev/stamp_awp_tail.sh): thread 0 of workgroup 0 sums the shader-clock cycles between the phase barriers into the tail
of the workspace.   EVD_LIB_PATH=evdeblurnerf_amd/lib/variants/libevd_atstamp.so python tools/dev/stamp_awp_tail.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from evdeblurnerf_amd import _lib as L
from evdeblurnerf_amd.awp import FusedAWP
from awp_standin import RefLikeAWP

R, P, S, VF = 1024, 10, 128, 32
awp = RefLikeAWP(P=P, view_ch=VF, mam="corr").cuda().train()
fused = FusedAWP(awp)
ps = [t.detach().contiguous() for t in fused._tail_params()]
arr = (C.c_void_p * len(ps))(*[t.data_ptr() for t in ps])
desc = L.AwpTailDesc(P=P, S=S, VF=VF, dir_freqs=2, n_mot=2, training=1, bn_eps=1e-5, bn_momentum=0.1)
f32 = dict(dtype=torch.float32, device="cuda")
h, vf, rd = torch.randn((R, P, 64), **f32), torch.randn((R, VF), **f32), torch.randn((R * P, 3), **f32)
hi, hs = torch.randn((R, P, 64), **f32).relu(), torch.randn((R, S, 64), **f32).relu()
out, y, xg, stats = torch.empty((R, P), **f32), torch.empty((R, P, 32), **f32), torch.empty((R, P, 32), **f32), torch.empty((64,), **f32)
lib = L.lib()
rays = None if os.environ.get("RECOMPUTE") else torch.empty((R, lib.evd_awp_tail_saved_floats(C.byref(desc))), **f32)
nf, nb = lib.evd_awp_tail_workspace_bytes(C.byref(desc), R, 0), lib.evd_awp_tail_workspace_bytes(C.byref(desc), R, 1)
wf, wb = torch.zeros((nf,), dtype=torch.uint8, device="cuda"), torch.zeros((nb,), dtype=torch.uint8, device="cuda")
g = torch.randn((R, P), **f32)
d_h, d_vf, d_rd, d_hi, d_hs = torch.empty_like(h), torch.empty_like(vf), torch.empty_like(rd), torch.empty_like(hi), torch.empty_like(hs)
d_par = torch.empty((lib.evd_awp_tail_param_count(C.byref(desc)),), **f32)


def fwd():
    L.check(lib.evd_awp_tail_forward(C.byref(desc), arr, L.ptr(h), L.ptr(vf), L.ptr(rd), L.ptr(hi), L.ptr(hs), R, None, None, None, L.ptr(out), L.ptr(y),
                                     L.ptr(xg), L.ptr(stats), L.ptr(rays), L.ptr(wf), nf, L.stream_ptr()), "fwd")


def bwd():
    L.check(lib.evd_awp_tail_backward(C.byref(desc), arr, L.ptr(h), L.ptr(vf), L.ptr(rd), L.ptr(hi), L.ptr(hs), R, L.ptr(y), L.ptr(xg), L.ptr(stats), L.ptr(rays), L.ptr(g),
                                      L.ptr(d_h), L.ptr(d_vf), L.ptr(d_rd), L.ptr(d_hi), L.ptr(d_hs), L.ptr(d_par), L.ptr(wb), nb, L.stream_ptr()), "bwd")


for fn, name in ((fwd, "forward"), (bwd, "backward")):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 20 * 1e3:.0f} us per call ({R} rays, P {P}, S {S})")
FW = ["stage weights", "inputs -> LDS", "layer 0, MAM.linear on inter / intra sums", "layers 1.., conva", "-", "convc, convn, convb", "logits, convl", "softmaxes",
      "attention sums", "convd", "y / x_global / BatchNorm sums out"]
BW = ["BatchNorm backward", "convd", "d attention maps", "d nI, softmax backward", "d q, d kP, d kI, convn / convl gradients", "conva / convb / convc gradients, d li, d ls",
      "MAM.linear gradient, d h_inter, d h_intra", "ReLU masks", "motion embedding layers", "d h, d view", "d rays_d"]
rays = (R + 255) // 256
for ws, n, names, which in ((wf, nf, FW, "forward"), (wb, nb, FW + BW, "backward")):
    t = ws[n - 512:n - 512 + 256].view(torch.int64).cpu().numpy().astype(np.float64)
    if not t.any():
        raise SystemExit("library was not built with -DEVD_AT_STAMP")
    print(f"{which}: workgroup 0, {rays} rays; shader-clock cycles per ray (100 MHz counter x ~24 = core cycles)")
    for i, nm in enumerate(names):
        if nm != "-":
            print(f"   {t[i] / (1 if i == 0 else rays):10.0f}  {nm}")
    print(f"   {t[1:].sum() / rays:10.0f}  total per ray")
