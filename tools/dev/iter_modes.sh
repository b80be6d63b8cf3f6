#!/bin/bash
# developer script: the blurfactory training iteration in every training mode, same box
for p in f16 f16c f16m f16x3; do python tools/bench_train_step.py --precision $p --iters 20 2>&1 | tail -1; done
for p in f16 f16c f16m; do python tools/bench_train_step.py --precision $p --iters 10 --awp fused 2>&1 | tail -1; done
