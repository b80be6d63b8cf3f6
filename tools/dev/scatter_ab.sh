#!/bin/bash
# developer A/B on one box: the tri-plane scatter with the basis gradient inside the main kernel (default) vs round 3's separate form
for sl in 0 0.05 0.35; do
  python tools/bench_voxel_bwd.py --slope $sl 2>&1 | tail -1 | sed 's/.*with scratch/with scratch/' | cut -c1-160
  python tools/bench_voxel_bwd.py --slope $sl --dpts 2>&1 | tail -1 | sed 's/.*with scratch/   dpts: with scratch/' | cut -c1-160
  EVD_SCATTER_BASIS=separate python tools/bench_voxel_bwd.py --slope $sl 2>&1 | tail -1 | sed 's/.*with scratch/   separate: with scratch/' | cut -c1-160
  EVD_SCATTER_BASIS=separate python tools/bench_voxel_bwd.py --slope $sl --dpts 2>&1 | tail -1 | sed 's/.*with scratch/   separate dpts: with scratch/' | cut -c1-160
done
python tools/bench_train_step.py --iters 20 2>&1 | tail -1
EVD_SCATTER_BASIS=separate python tools/bench_train_step.py --iters 20 2>&1 | tail -1 | sed 's/^/  separate: /'
