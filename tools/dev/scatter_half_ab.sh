#!/bin/bash
# developer A/B (GPU box): the scatter's re-gather on the float16 grid copies (default in the modes whose forward gathers them) against the
# float32 grids (EVD_SCATTER_HALF=0): tests, the iteration by mode, the scatter kernels' times   -> gpurun_out/r06_scatter_half_ab.log
out=gpurun_out/r06_scatter_half_ab.log; : > $out
python -m pytest tests/test_gpu_train.py tests/test_gpu_train_call.py tests/test_gpu_train_f32grade.py tests/test_gpu_train_f16c.py tests/test_gpu_fullsize.py tests/test_gpu_train_engine.py tests/test_gpu_bwd_fusion.py -q -x --tb=short 2>&1 | tail -4 >> $out
for r in 1 2 3; do for h in 1 0; do for p in f16 f16c; do
  echo "== EVD_SCATTER_HALF=$h $p (round $r)" >> $out
  EVD_SCATTER_HALF=$h python tools/bench_train_step.py --precision $p --iters 20 2>&1 | tail -1 >> $out
done; done; done
for h in 1 0; do echo "== kernels, EVD_SCATTER_HALF=$h" >> $out; EVD_SCATTER_HALF=$h python tools/profile_train_kernels.py 2>&1 | grep -E "iteration|sample_bwd|scatter_lines" >> $out; done
