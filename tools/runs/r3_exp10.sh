#!/bin/bash
O=gpurun_out/r3j; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/gputest.log 2>&1; tail -4 $O/gputest.log
cd /tmp && export TMPDIR=/tmp
for p in f16c f16; do
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$p -- python $GRAFT_REPO_ROOT/tools/bench_c2f.py --precision $p --iters 50 > /dev/null 2>&1
f=$(ls $GRAFT_REPO_ROOT/$O/prof_$p/*/*kernel_stats.csv | head -1); cp $f $GRAFT_REPO_ROOT/$O/c2f_${p}_kernel_stats.csv; rm -rf $GRAFT_REPO_ROOT/$O/prof_$p; head -6 $GRAFT_REPO_ROOT/$O/c2f_${p}_kernel_stats.csv | cut -c1-150
done
