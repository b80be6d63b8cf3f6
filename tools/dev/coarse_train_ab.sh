#!/bin/bash
# developer A/B (GPU box): the 64-wide level's TRAINING forward on the resident stream (default) against k_voxel_mlp_pipe's TRAIN variant (EVD_COARSE_FORM=pipe)
out=gpurun_out/r06_coarse_train_ab.log; : > $out
python -m pytest tests/test_gpu_train.py tests/test_gpu_train_call.py tests/test_gpu_train_f32grade.py tests/test_gpu_train_f16c.py tests/test_gpu_fullsize.py tests/test_gpu_train_engine.py tests/test_gpu_bwd_fusion.py tests/test_gpu_c2f_trained.py -q -x --tb=short 2>&1 | tail -3 >> $out
for r in 1 2 3; do for form in resident pipe; do for p in f16 f16c f16m; do
  unset EVD_COARSE_FORM; [ $form = pipe ] && export EVD_COARSE_FORM=pipe
  echo "== $form $p (round $r)" >> $out
  python tools/bench_train_step.py --precision $p --iters 20 2>&1 | tail -1 >> $out
done; done; done
for form in resident pipe; do unset EVD_COARSE_FORM; [ $form = pipe ] && export EVD_COARSE_FORM=pipe
  echo "== kernels, $form" >> $out; python tools/profile_train_kernels.py 2>&1 | grep -E "iteration|64, 15, 32" >> $out; done
