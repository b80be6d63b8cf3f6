#!/bin/bash
# round 3, GPU call 2: the compensated float16 mode of the PDRF fine level -- new tests, smoke, c2f render timing per mode, kernel trace
O=gpurun_out/r3b; mkdir -p $O
python -m pytest tests/test_gpu_c2f_trained.py -x -q -m gpu -s > $O/test_c2f_trained.log 2>&1; tail -15 $O/test_c2f_trained.log
python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -s -k "blurfactory" > $O/test_fullsize.log 2>&1; tail -5 $O/test_fullsize.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -12 $O/smoke.log
for p in f16 f16c f16x3; do python tools/bench_c2f.py --precision $p --iters 50 2>&1 | tail -1; done | tee $O/bench_c2f.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/tools/bench_c2f.py --precision f16c --iters 50 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls $O/prof/*/*kernel_stats.csv | head -1); cp $f $O/c2f_f16c_kernel_stats.csv; rm -rf $O/prof; head -14 $O/c2f_f16c_kernel_stats.csv | cut -c1-200
