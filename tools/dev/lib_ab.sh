#!/bin/bash
# developer A/B (GPU box) of the whole training iteration: the library in the tree against variant builds (tools/build_variant.sh)
#   bash tools/dev/lib_ab.sh <variant name> [precision]     -> gpurun_out/r06_lib_ab_<variant>.log
v=$1; prec=${2:-f16}; out=gpurun_out/r06_lib_ab_$v.log; : > $out
for r in 1 2 3; do for lib in default $v; do
  if [ $lib = default ]; then unset EVD_LIB_PATH; else export EVD_LIB_PATH=$PWD/evdeblurnerf_amd/lib/variants/libevd_$lib.so; fi
  echo "== $lib (round $r)" >> $out
  python tools/bench_train_step.py --precision $prec --iters 20 2>&1 | tail -1 >> $out
done; done
