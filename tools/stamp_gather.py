#!/usr/bin/env python
"""Phase stamps of the wavefront-autonomous gather k_voxel_sample_w (kernel_voxel.hip VS_STAMP, a library built with -DEVD_VS_TRACE):
cycles between the stamps of wavefront 0 of the first 8192 blocks, at the blurfactory fine level.  GPU box only.
    EVD_LIB_PATH=evdeblurnerf_amd/lib/variants/libevd_vstrace.so python tools/stamp_gather.py [--precision f16c]"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from evdeblurnerf_amd import _lib as L  # noqa: E402
from evdeblurnerf_amd import weights as W  # noqa: E402
from evdeblurnerf_amd.renderer import NeRFAll  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="f16c")
ap.add_argument("--samples", type=int, default=1 << 19)
a = ap.parse_args()
sd = W.make_blurfactory_state_dict(31)
model = NeRFAll(W.blurfactory_args(64), sd, precision=a.precision).eval()
fine = model.mlp_fine
rs = np.random.RandomState(3)
R, S = a.samples // 128, 128
# NDC-like rays: x, y nearly constant along a ray, z ascending
o = rs.uniform(-1.2, 1.2, size=(R, 1, 2)).astype(np.float32)
sl = rs.uniform(-0.05, 0.05, size=(R, 1, 2)).astype(np.float32)
z = np.sort(rs.uniform(-0.95, 0.95, size=(R, S, 1)).astype(np.float32), axis=1)
pts = torch.as_tensor(np.concatenate([o + sl * z, z], -1), device="cuda")
for _ in range(3):
    f = fine.sample(pts)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    f = fine.sample(pts)
e1.record()
e1.synchronize()
print(f"gather of {R * S} samples: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
buf = np.zeros(8 * 8192, dtype=np.int64)
fn = L.lib().evd_debug_vs_trace              # (lib() is the CDLL handle)
fn.argtypes = [C.c_void_p]
rc = fn(buf.ctypes.data_as(C.c_void_p))
t = buf.reshape(8192, 8)
d = np.diff(t[:, :5], axis=1).astype(np.float64)
names = ["basis issue + geometry + barrier (0->1)", "gather + finish -> coef (1->2)", "basis GEMM (2->3)", "transpose + store (3->4)"]
print("cycles per phase (wavefront 0 of a block, 16 samples; REALTIME counter at 100 MHz -> x 24 for shader cycles):")
for k, nme in enumerate(names):
    print(f"  {nme:45s} median {np.median(d[:, k]):8.0f}  mean {d[:, k].mean():8.0f}")
print(f"  total 0->4 median {np.median(t[:, 4] - t[:, 0]):.0f}")
print(f"  span of the launch (first stamp 0 to last stamp 4): {t[:, 4].max() - t[:, 0].min()}")
