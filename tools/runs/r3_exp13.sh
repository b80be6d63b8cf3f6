#!/bin/bash
O=gpurun_out/r3m; mkdir -p $O
for occ in 2 3; do
  for i in 1 2; do EVD_VW_F32_OCC=$occ python tools/bench_c2f.py --precision f16c --iters 50 2>&1 | tail -1; done
  EVD_VW_F32_OCC=$occ python tools/bench_c2f.py --precision f16x3 --iters 50 2>&1 | tail -1
done | tee $O/c2f_occ.log
python tools/bench_train_step.py --iters 20 2>&1 | tail -1 | tee $O/train_step.log
EVD_VW_F32_OCC=2 python tools/bench_train_step.py --iters 20 2>&1 | tail -1 | tee -a $O/train_step.log
python -m pytest tests/test_gpu_c2f_trained.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -2
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "loss or crf or event" 2>&1 | tail -2
