#!/bin/bash
# developer check (GPU box): TV loss / gradient as one launch per level (round 6): the tests that cover it, the iteration, the TV kernels' rows of the per-kernel table
out=gpurun_out/r06_tv_level.log; : > $out
python -m pytest tests/test_gpu_train.py tests/test_gpu_train_call.py tests/test_gpu_train_f32grade.py tests/test_gpu_train_engine.py tests/test_gpu_fullsize.py -q -x 2>&1 | tail -1 >> $out
for r in 1 2 3; do python tools/bench_train_step.py --precision f16 --iters 20 2>&1 | tail -1 >> $out; done
python tools/profile_train_kernels.py 2>&1 | grep -E "iteration|k_tv" >> $out
