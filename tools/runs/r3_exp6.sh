#!/bin/bash
# round 3, GPU call 6: per-ray AWP tail as a captured graph; basis-grad kernel at full occupancy
O=gpurun_out/r3f; mkdir -p $O
python -m pytest tests/test_gpu_awp.py -x -q -m gpu > $O/test_awp.log 2>&1; tail -12 $O/test_awp.log
python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "scatter or sample or voxel" > $O/test_scatter.log 2>&1; tail -3 $O/test_scatter.log
for slope in 0.0 0.05 0.35; do python tools/bench_voxel_bwd.py --slope $slope 2>&1 | tail -1; done | tee $O/scatter.log
python tools/bench_train_step.py --iters 20 2>&1 | tail -1 | tee $O/train_step.log
python tools/bench_train_step.py --iters 10 --awp fused 2>&1 | tail -1 | tee -a $O/train_step.log
python tools/bench_train_step.py --iters 10 --awp fused --no-graph 2>&1 | tail -1 | tee -a $O/train_step.log
python tools/bench_mam.py 2>&1 | tail -3 | tee $O/bench_mam.log
