#!/bin/bash
# developer check (GPU box): k_voxel_mlp_resident against k_voxel_mlp_pipe (EVD_COARSE_FORM=pipe) through evd_voxel_forward with feature = NULL: bit equality + time
out=gpurun_out/r06_coarse_form_eq.log; : > $out
for p in f16x3 f16 bf16; do
  python tools/dev/coarse_form_check.py /tmp/res_$p.npy $p 2>&1 | grep "level forward\|Error\|error" >> $out
  EVD_COARSE_FORM=pipe python tools/dev/coarse_form_check.py /tmp/pipe_$p.npy $p 2>&1 | grep "level forward\|Error\|error" >> $out
  python -c "import numpy as np; a=np.load('/tmp/res_$p.npy'); b=np.load('/tmp/pipe_$p.npy'); print('$p: resident vs pipe: max |diff| =', float(np.abs(a-b).max()), 'over', a.size, 'values; bit-equal:', bool((a.view(np.uint32)==b.view(np.uint32)).all()))" >> $out
done
