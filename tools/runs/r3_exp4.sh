#!/bin/bash
# round 3, GPU call 4: whole GPU suite, the bench line (with the c2f parity block), the N = 2 path on one GPU, the remaining torch glue by call site
O=gpurun_out/r3d; mkdir -p $O
python -m pytest tests -q -m gpu -x > $O/gputest.log 2>&1; tail -4 $O/gputest.log
( time python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real; tail -c 600 $O/bench.json; tail -3 $O/bench.err
EVD_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 > $O/bench_gpus2_shared.json 2> $O/bench_gpus2_shared.err; tail -c 1500 $O/bench_gpus2_shared.json; tail -3 $O/bench_gpus2_shared.err
python tools/profile_train_ops.py > $O/profile_train_ops.txt 2>&1; head -50 $O/profile_train_ops.txt
python tools/bench_train_step.py --iters 20 2>&1 | tail -1 | tee $O/train_step.log
