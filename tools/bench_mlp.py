#!/usr/bin/env python
"""Kernel-level timing of the fused MLP kernels (HIP events), all precisions. GPU box only.
    python tools/bench_mlp.py [--rays 4096] [--samples 128] [--iters 30]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

from evdeblurnerf_amd import _lib as L, weights as W  # noqa: E402
from evdeblurnerf_amd.nerf import NeRF  # noqa: E402

FLOP = 2 * 593408


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--precs", default="f32,f16x3,bf16,f16")
    ap.add_argument("--zero", action="store_true", help="all-zero weights and rays: same instruction stream, minimal switching power (DVFS evidence)")
    a = ap.parse_args()
    R, S = a.rays, a.samples
    sd = W.make_nerf_state_dict(21)
    if a.zero:
        sd = {k: np.zeros_like(v) for k, v in sd.items()}
    net = NeRF(sd)
    rs = np.random.RandomState(0)
    rb = np.zeros((R, 11), np.float32)
    rb[:, :3] = rs.uniform(-1, 1, (R, 3))
    rb[:, 3:6] = rs.uniform(-1, 1, (R, 3))
    rb[:, 7] = 1
    vd = rs.standard_normal((R, 3))
    rb[:, 8:] = vd / np.linalg.norm(vd, axis=-1, keepdims=True)
    if a.zero:
        rb[:] = 0
    rb = torch.as_tensor(rb, device="cuda")
    z = torch.linspace(0, 1, S, device="cuda").expand(R, S).contiguous()
    ref = None
    for prec in a.precs.split(","):
        it = max(3, a.iters // (6 if prec == "f32" else 1))
        for _ in range(3):
            raw, _ = net.mlpforward(rb, z, precision=prec)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(it):
            raw, _ = net.mlpforward(rb, z, precision=prec)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / it
        tf = R * S * FLOP / (ms * 1e-3) / 1e12
        if ref is None:
            ref = raw.clone() if prec == "f32" else net.mlpforward(rb, z, precision="f32")[0].clone()
        err = float((raw - ref).abs().max())
        print(f"{prec:6s} {ms:8.3f} ms  {tf:7.1f} TFLOP/s (algorithmic)  {R / (ms * 1e-3) / 1e6:6.2f} M rays/s   max|raw - f32| = {err:.2e}")


if __name__ == "__main__":
    main()
