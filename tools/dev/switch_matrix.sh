#!/bin/bash
# developer check (GPU box): the documented developer switches of this round still select working code: a subset of the GPU suite under each
out=gpurun_out/r06_switch_matrix.log; : > $out
T="tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_train.py tests/test_gpu_train_call.py"
for sw in "EVD_COARSE_FORM=pipe" "EVD_COARSE_TRIG=exact" "EVD_SCATTER_HALF=0" "EVD_SCATTER_LINES_ORDER=old" "EVD_SCATTER_LINES_ORDER=job" "EVD_GATHER_FORM=w" "EVD_SCATTER_ISSUER=1" "EVD_BWD_PIPE=0"; do
  echo "== $sw" >> $out
  env $sw python -m pytest $T -q -x 2>&1 | tail -1 >> $out
done
