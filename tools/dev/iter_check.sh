# developer check (GPU box) after a change to a training kernel: the tests that cover it, the iteration by mode, the per-kernel table
#   bash tools/dev/iter_check.sh <tag>  -> gpurun_out/r06_<tag>_{tests.log,iter.log,train_kernels.txt}
tag=${1:-check}
python -m pytest tests/test_gpu_train.py tests/test_gpu_train_call.py tests/test_gpu_train_f32grade.py tests/test_gpu_train_f16c.py tests/test_gpu_fullsize.py tests/test_gpu_train_engine.py tests/test_gpu_bwd_fusion.py -q -x --tb=short 2>&1 | tail -5 > gpurun_out/r06_${tag}_tests.log
for p in f16 f16c f16m; do python tools/bench_train_step.py --precision $p --iters 20 2>&1 | tail -1; done > gpurun_out/r06_${tag}_iter.log
python tools/profile_train_kernels.py > gpurun_out/r06_${tag}_train_kernels.txt 2>&1
