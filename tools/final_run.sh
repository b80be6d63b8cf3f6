mkdir -p gpurun_out/s5/final
python -m pytest tests -m gpu -q > gpurun_out/s5/final/gputest.log 2>&1; tail -3 gpurun_out/s5/final/gputest.log
python bench.py > gpurun_out/s5/final/bench.json 2> gpurun_out/s5/final/bench.err; tail -c 300 gpurun_out/s5/final/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/s5/final/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-train --no-parity --no-c2f --no-strong > $GRAFT_REPO_ROOT/gpurun_out/s5/final/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/s5/final/rocprof.err
cd $GRAFT_REPO_ROOT
f=$(ls gpurun_out/s5/final/prof/*/*kernel_stats.csv | head -1); cp $f gpurun_out/s5/final/bench_kernel_stats.csv; head -4 $f | cut -c1-160
