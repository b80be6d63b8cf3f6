# A/B of whole-library builds (GPU box): bash tools/run_lib_variants.sh <lib.so> <lib.so> ...   (paths relative to the repo root)
for r in 1 2; do
for v in "$@"; do
  echo "==== $v (round $r)"; export EVD_LIB_PATH=$PWD/$v
  timeout 200 python tools/bench_mlp.py --precs f16x3,bf16,f16,f16c --iters 60 2>&1 | grep -E "^(f16|bf16|f32)"
  timeout 200 python tools/bench_train.py --precs f16,f16x3 2>&1 | grep -v amdgpu.ids | tail -6
  timeout 200 python tools/bench_c2f.py 2>&1 | grep -v amdgpu.ids | tail -2
  timeout 300 python tools/bench_train_step.py --iters 10 2>&1 | grep -v amdgpu.ids | tail -1
done; done
