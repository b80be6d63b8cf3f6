#!/bin/bash
# developer A/B (GPU box): the scatter with phase 1 on split-float16 MFMAs (tree) against the float32 16x16x4 form (variant kvhead = HEAD's kernel_voxel.o)
#   bash tools/dev/scatter_phase1_ab.sh -> gpurun_out/r06_scatter_phase1_ab.log
out=gpurun_out/r06_scatter_phase1_ab.log; : > $out
timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_call.py tests/test_gpu_voxnerf.py -x -q 2>&1 | tail -3 >> $out
for r in 1 2 3; do for lib in default kvhead; do
  if [ $lib = default ]; then unset EVD_LIB_PATH; else export EVD_LIB_PATH=$PWD/evdeblurnerf_amd/lib/variants/libevd_$lib.so; fi
  echo "== $lib (round $r)" >> $out
  python tools/bench_voxel_bwd.py 2>&1 | grep -iE "scatter|backward" | head -4 >> $out
  python tools/bench_voxel_bwd.py --dpts 2>&1 | grep -iE "scatter|backward" | head -4 >> $out
  python tools/bench_train_step.py --precision f16 --iters 20 2>&1 | tail -1 >> $out
done; done
