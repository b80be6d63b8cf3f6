#!/bin/bash
# round 3, GPU call 9: coarse level on the software pipeline in inference; whole suite
O=gpurun_out/r3i; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/gputest.log 2>&1; tail -4 $O/gputest.log
for p in f16 f16c f16x3; do python tools/bench_c2f.py --precision $p --iters 50 2>&1 | tail -1; done | tee $O/bench_c2f.log
