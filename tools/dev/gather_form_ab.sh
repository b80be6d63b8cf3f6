#!/bin/bash
# developer A/B (GPU box): the forward gather k_voxel_sample_m (round 6) against k_voxel_sample_w (EVD_GATHER_FORM=w): parity tests, then kernel times
cd /tmp && export TMPDIR=/tmp; cd - > /dev/null
out=gpurun_out/r06_gather_form_ab.log; : > $out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_c2f_trained.py -q -x --tb=short 2>&1 | tail -4 >> $out
for form in m w; do
  export EVD_GATHER_FORM=$form
  for prec in f16 f16c f16x3; do
    rm -rf gpurun_out/gf_prof
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gf_prof -- python tools/bench_c2f.py --precision $prec > gpurun_out/gf_run.log 2>&1
    echo "== EVD_GATHER_FORM=$form precision $prec" >> $out
    grep -E "c2f|render" gpurun_out/gf_run.log | tail -1 | cut -c1-220 >> $out
    f=$(find gpurun_out/gf_prof -name "*kernel_stats.csv" | head -1)
    python - "$f" >> $out <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_voxel_sample_" in r["Name"]:
        print(f"   {r['Name'][:60]:60s} calls {r['Calls']:>5s}  average {float(r['AverageNs'])/1e3:8.1f} us")
PY
  done
done
cat $out
