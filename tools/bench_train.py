#!/usr/bin/env python
"""Timing of the NeRF-mode training kernels: forward that keeps activations, backward of the fused MLP. GPU box only.
    python tools/bench_train.py [--rays 4096] [--samples 128] [--iters 20] [--precs f16,bf16]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from evdeblurnerf_amd import weights as W  # noqa: E402
from evdeblurnerf_amd.nerf import NeRF  # noqa: E402

FLOP = 2 * 593408


def timed(fn, iters):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--precs", default="f16,bf16")
    a = ap.parse_args()
    R, S = a.rays, a.samples
    net = NeRF(W.make_nerf_state_dict(21))
    rs = np.random.RandomState(0)
    rb = np.zeros((R, 11), np.float32)
    rb[:, :3] = rs.uniform(-1, 1, (R, 3))
    rb[:, 3:6] = rs.uniform(-1, 1, (R, 3))
    rb[:, 7] = 1
    vd = rs.standard_normal((R, 3))
    rb[:, 8:] = vd / np.linalg.norm(vd, axis=-1, keepdims=True)
    rb = torch.as_tensor(rb, device="cuda")
    z = torch.linspace(0, 1, S, device="cuda").expand(R, S).contiguous()
    d_raw = torch.randn((R, S, 4), device="cuda") * 1e-4
    for prec in a.precs.split(","):
        fi = timed(lambda: net.mlpforward(rb, z, precision=prec), a.iters)
        ft = timed(lambda: net.mlpforward_train(rb, z, precision=prec), a.iters)
        raw, store = net.mlpforward_train(rb, z, precision=prec)
        bw = timed(lambda: net.mlp_backward(d_raw, store, precision=prec), a.iters)
        tf = lambda ms, mult: R * S * FLOP * mult / (ms * 1e-3) / 1e12
        print(f"{prec:5s} R={R} S={S}: inference fwd {fi:.3f} ms ({tf(fi, 1):.0f} TF/s) | training fwd {ft:.3f} ms | "
              f"backward {bw:.3f} ms ({tf(bw, 2):.0f} TF/s of 2x fwd flops) | store {store.numel() / 2**30:.2f} GiB")


if __name__ == "__main__":
    main()
