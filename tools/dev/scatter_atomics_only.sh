#!/bin/bash
# developer A/B (GPU box): the tri-plane scatter's main kernel against a build of it that keeps only the tap geometry and the atomics
# (-DEVD_VBW_ATOMICS_ONLY: constant rows, no gather, no MFMA): what the REAL address stream's atomics cost alone.
#   bash tools/build_variant.sh vbw_ao kernel_voxel.hip -DEVD_VBW_ATOMICS_ONLY     (build container)
#   bash tools/dev/scatter_atomics_only.sh                                          (GPU box) -> gpurun_out/r06_scatter_atomics_only.log
cd /tmp && export TMPDIR=/tmp; cd - > /dev/null
out=gpurun_out/r06_scatter_atomics_only.log; : > $out
for slope in 0.05; do
for v in default vbw_ao vbw_na; do
  if [ $v = default ]; then unset EVD_LIB_PATH; else export EVD_LIB_PATH=$PWD/evdeblurnerf_amd/lib/variants/libevd_$v.so; fi
  rm -rf gpurun_out/ao_prof
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ao_prof -- python tools/bench_voxel_bwd.py --slope $slope --iters 20 > gpurun_out/ao_run.log 2>&1
  echo "== slope $slope, library $v" >> $out
  grep -h "scatter backward" gpurun_out/ao_run.log | sed 's/.*| scatter backward/scatter backward/' >> $out
  f=$(find gpurun_out/ao_prof -name "*kernel_stats.csv" | head -1); ls -R gpurun_out/ao_prof > gpurun_out/ao_ls.log
  python - "$f" >> $out <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if "k_voxel_sample_bwd_w" in r["Name"] or "k_scatter_lines" in r["Name"]:
        print(f"   {r['Name'][:60]:60s} calls {r['Calls']:>5s}  average {float(r['AverageNs'])/1e3:8.1f} us")
PY
done; done
cat $out
