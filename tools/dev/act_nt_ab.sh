#!/bin/bash
# developer A/B (GPU box): the fine level's training forward with non-temporal fragment stores (-DEVD_ACT_NT on kernel_voxel_train_f16.hip)
out=gpurun_out/r06_act_nt_ab.log; : > $out
for r in 1 2 3; do for v in default vtrain_nt; do
  if [ $v = default ]; then unset EVD_LIB_PATH; else export EVD_LIB_PATH=$PWD/evdeblurnerf_amd/lib/variants/libevd_$v.so; fi
  echo "== $v (round $r)" >> $out
  python tools/bench_train_step.py --precision f16 --iters 20 2>&1 | tail -1 >> $out
done; done
unset EVD_LIB_PATH
python tools/profile_train_kernels.py 2>&1 | grep -E "iteration|k_voxel_mlp_pipe|k_wgrad_dgrad|k_voxel_bwd_fused64" | cut -c1-140 >> $out
export EVD_LIB_PATH=$PWD/evdeblurnerf_amd/lib/variants/libevd_vtrain_nt.so
echo "== kernels with vtrain_nt" >> $out
python tools/profile_train_kernels.py 2>&1 | grep -E "iteration|k_voxel_mlp_pipe|k_wgrad_dgrad|k_voxel_bwd_fused64" | cut -c1-140 >> $out
