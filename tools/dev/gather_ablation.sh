#!/bin/bash
# developer ablation (GPU box): the forward tri-plane gather k_voxel_sample_w with and without its grid loads (-DEVD_VS_NO_LOADS)
cd /tmp && export TMPDIR=/tmp; cd - > /dev/null
out=gpurun_out/r06_gather_ablation.log; : > $out
for v in default vs_noloads; do
  if [ $v = default ]; then unset EVD_LIB_PATH; else export EVD_LIB_PATH=$PWD/evdeblurnerf_amd/lib/variants/libevd_$v.so; fi
  rm -rf gpurun_out/ga_prof
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ga_prof -- python tools/bench_c2f.py > gpurun_out/ga_run.log 2>&1
  echo "== library $v" >> $out
  tail -2 gpurun_out/ga_run.log | cut -c1-200 >> $out
  f=$(find gpurun_out/ga_prof -name "*kernel_stats.csv" | head -1)
  python - "$f" >> $out <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_voxel_sample_w" in r["Name"] or "k_voxel_mlp" in r["Name"]:
        print(f"   {r['Name'][:70]:70s} calls {r['Calls']:>5s}  average {float(r['AverageNs'])/1e3:8.1f} us")
PY
done
cat $out
