#!/usr/bin/env python
"""Does the compensated mode (f16c) hold the 1e-4 RGB bound with the float16 copies of the tri-plane grids instead of the float32 ones?
Trains the blurfactory-size c2f model once (tools/trained_c2f.py), then measures RGB L-inf vs the oracle and the render time with
EVD_F16C_HALF_GRIDS = 0 (float32 grids on both levels), 1 (coarse level float16), 2 (fine level: the shipped choice), 3 (both), 4 (the coarse level at the importance samples only), 6 (2 + 4).  GPU box only.
    python tests/dev_f16c_grid_modes.py [--iters 3000]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import torch  # noqa: E402

import trained_c2f as TC  # noqa: E402
from oracle import oracle as O  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=3000)
a = ap.parse_args()
O.build()
sd, rep = TC.train_c2f(iters=a.iters)
print("trained:", {k: rep[k] for k in ("iters", "loss_first", "loss_last")})
for m in ("0", "1", "2", "3", "4", "6"):
    os.environ["EVD_F16C_HALF_GRIDS"] = m
    err, info = TC.c2f_parity(O, sd, ("f16c",))
    print(f"EVD_F16C_HALF_GRIDS={m}: RGB L-inf vs oracle, trained, 4096 x (64 + 64): fine {err['f16c']['fine']:.2e} coarse {err['f16c']['coarse']:.2e}")
