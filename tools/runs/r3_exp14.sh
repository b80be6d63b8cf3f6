#!/bin/bash
O=gpurun_out/r3n; mkdir -p $O
EVD_LIB_PATH=$PWD/evdeblurnerf_amd/lib/variants/libevd_vstrace.so python tools/stamp_gather.py 2>&1 | tail -8 | tee $O/stamps.log
python tools/stamp_gather.py 2>&1 | grep "gather of" | tee -a $O/stamps.log
for i in 1 2; do python tools/bench_c2f.py --precision f16c --iters 50 2>&1 | tail -1; done | tee $O/c2f.log
python tools/bench_c2f.py --precision f16 --iters 50 2>&1 | tail -1 | tee -a $O/c2f.log
python -m pytest tests/test_gpu_c2f_trained.py tests/test_gpu_fullsize.py tests/test_gpu_voxel.py -x -q -m gpu 2>&1 | tail -2
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "c2f or appfeature or voxel or sample" 2>&1 | tail -2
