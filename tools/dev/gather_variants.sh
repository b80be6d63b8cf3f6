#!/bin/bash
# developer A/B (GPU box): variant builds of the forward gather (tools/build_variant.sh vm_* kernel_voxel.hip ...): kernel time under rocprofv3
cd /tmp && export TMPDIR=/tmp; cd - > /dev/null
out=gpurun_out/r06_gather_variants.log; : > $out
for v in default "$@"; do
  if [ $v = default ]; then unset EVD_LIB_PATH; else export EVD_LIB_PATH=$PWD/evdeblurnerf_amd/lib/variants/libevd_$v.so; fi
  for prec in f16 f16c; do
    rm -rf gpurun_out/gv_prof
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gv_prof -- python tools/bench_c2f.py --precision $prec > gpurun_out/gv_run.log 2>&1
    echo "== $v precision $prec: $(grep -E 'ms' gpurun_out/gv_run.log | tail -1 | cut -c1-160)" >> $out
    f=$(find gpurun_out/gv_prof -name "*kernel_stats.csv" | head -1)
    python - "$f" >> $out <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_voxel_sample_" in r["Name"]:
        print(f"   {r['Name'][:60]:60s} calls {r['Calls']:>5s}  average {float(r['AverageNs'])/1e3:8.1f} us")
PY
  done
done
cat $out
