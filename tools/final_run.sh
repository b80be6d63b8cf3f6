#!/bin/bash
# the round's closing GPU run: whole GPU suite, the bench line, the same bench under rocprofv3 (kernel stats), the training iteration under rocprofv3
O=${1:-gpurun_out/r4_final}; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; tail -3 $O/gputest.log
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-train --no-parity --no-c2f --no-strong --no-zero-probe > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/rocprof.err
cd $GRAFT_REPO_ROOT
f=$(ls $O/prof/*/*kernel_stats.csv | head -1); cp $f $O/bench_kernel_stats.csv; rm -rf $O/prof; head -4 $O/bench_kernel_stats.csv | cut -c1-160
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof2 -- python $GRAFT_REPO_ROOT/tools/bench_train_step.py --iters 8 > $GRAFT_REPO_ROOT/$O/train_step_rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls $O/prof2/*/*kernel_stats.csv | head -1); cp $f $O/train_step_kernel_stats.csv; rm -rf $O/prof2; head -6 $O/train_step_kernel_stats.csv | cut -c1-160
python tools/profile_train_kernels.py --top 70 2>&1 | grep -v "Warn\|amdgpu.ids\|_warn_once" > $O/train_kernels_steady.txt; head -5 $O/train_kernels_steady.txt
python tools/profile_train_kernels.py --awp fused --top 80 2>&1 | grep -v "Warn\|amdgpu.ids\|_warn_once" > $O/train_kernels_awp_steady.txt; head -5 $O/train_kernels_awp_steady.txt
for m in f16c f16m; do python tools/profile_train_kernels.py --precision $m --top 70 2>&1 | grep -v "Warn\|amdgpu.ids\|_warn_once" > $O/train_kernels_${m}_steady.txt; head -2 $O/train_kernels_${m}_steady.txt; done
python tools/bench_train_step.py --iters 20 2>&1 | tail -1 | tee $O/train_step.log
for m in f16c f16m; do python tools/bench_train_step.py --iters 20 --precision $m 2>&1 | tail -1 | tee -a $O/train_step.log; done
python tools/bench_train_step.py --iters 10 --awp fused 2>&1 | tail -1 | tee -a $O/train_step.log
