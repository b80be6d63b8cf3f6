# timing of f16c kernel variants built into evdeblurnerf_amd/lib/abl/ (GPU box):  bash tools/run_c_variants.sh base vf ...
for r in 1 2; do
for v in "$@"; do
  echo "== $v (round $r)"; EVD_LIB_PATH=$PWD/evdeblurnerf_amd/lib/abl/libevdnerf_$v.so timeout 120 python tools/bench_mlp.py --precs f16c --iters 100 2>&1 | grep f16c
done; done
