#!/bin/bash
# developer ablation (GPU box): what could hiding k_voxel_mlp_c's prologue work gain at most?  Variants built with -DEVD_VC_ABL=1 (the two
# positional encodings only on a workgroup's first pass) and =3 (the feature block too); results wrong by construction.
#   hipcc ... -DEVD_VC_ABL=<m> -c csrc/kernel_voxel_pipe_f16c.hip ; link as lib/variants/libevd_vcabl<m>.so ; bash tools/dev/voxel_c_prologue_ablation.sh
out=gpurun_out/r06_voxel_c_prologue_ablation.log; : > $out
for r in 1 2 3; do for lib in default vcabl1 vcabl3; do
  if [ $lib = default ]; then unset EVD_LIB_PATH; else export EVD_LIB_PATH=$PWD/evdeblurnerf_amd/lib/variants/libevd_$lib.so; fi
  echo "== $lib (round $r)" >> $out
  python tools/bench_c2f.py --precision f16c --iters 50 2>&1 | tail -1 >> $out
done; done
