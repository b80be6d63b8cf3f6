#!/bin/bash
O=gpurun_out/r3k; mkdir -p $O
python -m pytest tests/test_gpu_train.py tests/test_gpu_train_f32grade.py tests/test_gpu_train_engine.py -x -q -m gpu > $O/test_train.log 2>&1; tail -3 $O/test_train.log
for slope in 0.0 0.05 0.35; do python tools/bench_voxel_bwd.py --slope $slope 2>&1 | tail -1; done | tee $O/scatter.log
EVD_SCATTER_LINES_LDS=1 python tools/bench_voxel_bwd.py --slope 0.05 2>&1 | tail -1 | tee -a $O/scatter.log
python tools/bench_train_step.py --iters 20 2>&1 | tail -1 | tee $O/train_step.log
EVD_SCATTER_LINES_LDS=1 python tools/bench_train_step.py --iters 20 2>&1 | tail -1 | tee -a $O/train_step.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/tools/bench_train_step.py --iters 8 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls $O/prof/*/*kernel_stats.csv | head -1); cp $f $O/train_step_kernel_stats.csv; rm -rf $O/prof; head -8 $O/train_step_kernel_stats.csv | cut -c1-60,150-260
