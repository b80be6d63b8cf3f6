#!/bin/bash
# developer A/B (GPU box): workgroup order of k_scatter_lines: all jobs of a chunk on one XCD (default) / consecutive ids (job) / job-major as in rounds 3-5 (old)
out=gpurun_out/r06_scatter_lines_order_ab.log; : > $out
python -m pytest tests/test_gpu_train.py -q -x -k "scatter or triplane" 2>&1 | tail -1 >> $out
for o in xcd job old; do
  echo "== EVD_SCATTER_LINES_ORDER=$o" >> $out; EVD_SCATTER_LINES_ORDER=$o python tools/profile_train_kernels.py 2>&1 | grep -E "iteration|scatter_lines" >> $out
  for r in 1 2; do EVD_SCATTER_LINES_ORDER=$o python tools/bench_train_step.py --precision f16 --iters 20 2>&1 | tail -1 >> $out; done
done
