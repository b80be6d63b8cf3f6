#!/bin/bash
O=gpurun_out/r3q; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/tools/bench_train_step.py --iters 4 > $GRAFT_REPO_ROOT/$O/log.txt 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls $O/prof/*/*memory_copy_trace.csv | head -1); cp $f $O/memcopy.csv; rm -rf $O/prof; wc -l $O/memcopy.csv; head -3 $O/memcopy.csv
