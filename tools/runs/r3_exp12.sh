#!/bin/bash
O=gpurun_out/r3l; mkdir -p $O
python -m pytest tests/test_gpu_train.py tests/test_gpu_dist.py tests/test_gpu_train_engine.py -x -q -m gpu > $O/test_train.log 2>&1; tail -3 $O/test_train.log
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "loss or crf or event" > $O/test_loss.log 2>&1; tail -3 $O/test_loss.log
python tools/bench_train_step.py --iters 20 2>&1 | tail -1 | tee $O/train_step.log
python tools/trace_torch_ops.py > $O/trace_ops.txt 2>&1; tail -60 $O/trace_ops.txt
python tools/profile_train_ops.py > $O/profile_ops.txt 2>&1; grep -c . $O/profile_ops.txt
