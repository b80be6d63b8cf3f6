#!/usr/bin/env python
"""Time the REAL reference (imported from /root/reference, CPU, PyTorch) on the metric workload: NeRFAll.render,
mode=nerf 8x256, 4096 rays x 128 samples, perturb 0 -- the true-reference anchor of BASELINE.md (the reference cannot
travel to the GPU box, where bench.py's cpu_baseline is the parity-checked C restatement instead).  Build container only.
    python tools/time_reference_cpu.py [--rays 4096] [--reps 3]"""
import argparse
import contextlib
import io
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

ref_import.install()
import numpy as np  # noqa: E402
import torch  # noqa: E402

from evdeblurnerf_amd import weights as W  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    torch.set_grad_enabled(False)
    from networks.renderer import NeRFAll
    args = ref_import.blurfactory_args(mode="nerf", N_importance=0, rgb_add_bias=True)
    with contextlib.redirect_stdout(io.StringIO()):
        model = NeRFAll(args)
    ref_import.load_np_state_dict(model, W.prefixed(W.make_nerf_state_dict(21), "mlp_coarse"))
    model.train(False)
    K = torch.from_numpy(W.synthetic_camera())
    rays = torch.from_numpy(W.synthetic_rays(100, a.rays))
    kw = dict(ndc=True, near=0., far=1., use_viewdirs=True, N_samples=128, N_importance=0, retraw=False, perturb=0., raw_noise_std=0.)
    model.render(400, 400, K, 32768, rays=rays[:256], **kw)
    best = 1e9
    for _ in range(a.reps):
        t0 = time.perf_counter()
        rgb = model.render(400, 400, K, 32768, rays=rays, **kw)[0]
        best = min(best, time.perf_counter() - t0)
    print(f"reference NeRFAll.render (CPU, torch {torch.__version__}, {torch.get_num_threads()} threads): {a.rays} rays x 128 samples "
          f"in {best:.2f} s = {a.rays / best:.0f} rays/s   (rgb mean {float(rgb.mean()):.4f})")


if __name__ == "__main__":
    main()
