#!/bin/bash
# developer A/B on one box: a reference library (EVD_LIB_PATH=$1, "" = none) vs the current one: c2f render + training iteration + fine-level stamps
V=$1
for i in 1 2 3; do
  [ -n "$V" ] && EVD_LIB_PATH=$V python tools/bench_c2f.py --precision f16c --iters 50 2>&1 | tail -1 | sed 's/^/  ref: /'
  python tools/bench_c2f.py --precision f16c --iters 50 2>&1 | tail -1
done
[ -n "$V" ] && EVD_LIB_PATH=$V python tools/bench_train_step.py --precision f16c --iters 20 2>&1 | tail -1 | sed 's/^/  ref: /'
python tools/bench_train_step.py --precision f16c --iters 20 2>&1 | tail -1
EVD_LIB_PATH=evdeblurnerf_amd/lib/variants/libevd_vstamp.so python tools/dev/stamp_voxel_c.py 2>&1 | grep -v "Warn\|amdgpu.ids"
