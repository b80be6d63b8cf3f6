#!/bin/bash
# developer ablation (GPU box): k_scatter_lines without its LDS atomics (variant sl_na: -DEVD_SL_NO_ATOMICS) -- what its loads cost alone
out=gpurun_out/r06_scatter_lines_ablation.log; : > $out
for lib in default sl_na; do
  if [ $lib = default ]; then unset EVD_LIB_PATH; else export EVD_LIB_PATH=$PWD/evdeblurnerf_amd/lib/variants/libevd_$lib.so; fi
  echo "== $lib" >> $out; python tools/profile_train_kernels.py 2>&1 | grep -E "iteration|scatter_lines" >> $out; done
