python -m pytest tests/test_gpu_train.py tests/test_gpu_train_call.py tests/test_gpu_train_f32grade.py tests/test_gpu_fullsize.py tests/test_gpu_train_engine.py -q -x --tb=short 2>&1 | tail -5 > gpurun_out/r06_scatter_tests.log
for p in f16 f16c; do python tools/bench_train_step.py --precision $p --iters 20 2>&1 | tail -1; done > gpurun_out/r06_iter_after_walk.log
python tools/profile_train_kernels.py > gpurun_out/r06_train_kernels_after_walk.txt 2>&1
