#!/bin/bash
# round 3, GPU call 3: training-engine plumbing (flat parameter storage, in-place gradient accumulation, ray packing / points as kernels)
O=gpurun_out/r3c; mkdir -p $O
python -m pytest tests/test_gpu_train_engine.py -x -q -m gpu > $O/test_engine.log 2>&1; tail -15 $O/test_engine.log
python -m pytest tests/test_gpu_train.py tests/test_gpu_dist.py tests/test_gpu_awp.py tests/test_gpu_train_f32grade.py -x -q -m gpu > $O/test_train.log 2>&1; tail -8 $O/test_train.log
python tools/bench_train_step.py --iters 20 --plain-autograd 2>&1 | tail -1 | tee $O/train_step.log
python tools/bench_train_step.py --iters 20 2>&1 | tail -1 | tee -a $O/train_step.log
python tools/bench_train_step.py --iters 10 --awp fused 2>&1 | tail -1 | tee -a $O/train_step.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/tools/bench_train_step.py --iters 8 > $GRAFT_REPO_ROOT/$O/train_step_rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls $O/prof/*/*kernel_stats.csv | head -1); cp $f $O/train_step_kernel_stats.csv; rm -rf $O/prof
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r3c/train_step_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
ne=[r for r in rows if 'evd::' not in r['Name']]
print('iters: 3 warm-up + 8 timed = 11; total kernel ms/iter', tot/1e6/11, ' non-evd ms/iter', sum(float(r['TotalDurationNs']) for r in ne)/1e6/11, ' non-evd launches/iter', sum(int(r['Calls']) for r in ne)/11, ' evd launches/iter', sum(int(r['Calls']) for r in rows if 'evd::' in r['Name'])/11)
for r in sorted(ne,key=lambda r:-int(r['Calls']))[:25]:
    print(r['Calls'], f"{float(r['TotalDurationNs'])/1e6:.3f}", r['Name'][:130])
PY
