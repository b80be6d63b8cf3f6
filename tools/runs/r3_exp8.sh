#!/bin/bash
# round 3, GPU call 8: use_viewdirs=False networks (golden G23), AWP graph test, quick regression of the parity file
O=gpurun_out/r3h; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_awp.py -x -q -m gpu -s > $O/test_parity.log 2>&1; tail -6 $O/test_parity.log; grep "no viewdirs" $O/test_parity.log | head -12
