#!/bin/bash
# developer A/B (GPU box): the issuer-wavefront form of the scatter (EVD_SCATTER_ISSUER=1) against the default, both with the float16 re-gather, in the iteration
out=gpurun_out/r06_scatter_issuer_half_ab.log; : > $out
for r in 1 2 3; do for iss in 0 1; do
  echo "== EVD_SCATTER_ISSUER=$iss (round $r)" >> $out
  EVD_SCATTER_ISSUER=$iss python tools/bench_train_step.py --precision f16 --iters 20 2>&1 | tail -1 >> $out
done; done
for iss in 0 1; do echo "== kernels, EVD_SCATTER_ISSUER=$iss" >> $out; EVD_SCATTER_ISSUER=$iss python tools/profile_train_kernels.py 2>&1 | grep -E "iteration|sample_bwd|scatter_lines" >> $out; done
EVD_SCATTER_ISSUER=1 python -m pytest tests/test_gpu_train.py -q -x -k "scatter or triplane" 2>&1 | tail -2 >> $out
