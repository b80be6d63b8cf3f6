#!/bin/bash
O=gpurun_out/r3o; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/tools/bench_train_step.py --iters 10 --awp fused > $GRAFT_REPO_ROOT/$O/awp_rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls $O/prof/*/*kernel_stats.csv | head -1); cp $f $O/train_step_awp_kernel_stats.csv; rm -rf $O/prof; tail -1 $O/awp_rocprof.log
