cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for form in resident pipe; do
  O=$ROOT/gpurun_out/c2fprof_$form; rm -rf $O
  if [ $form = pipe ]; then export EVD_COARSE_FORM=pipe; else unset EVD_COARSE_FORM; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $ROOT/tools/bench_c2f.py --precision f16c --iters 50 > $O.log 2>&1
  python - <<PY >> $ROOT/gpurun_out/r06_c2f_kernels_$form.txt
import csv, glob
rows = []
for f in glob.glob("$O/**/*kernel_stats.csv", recursive=True): rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]: print(f"{float(r['TotalDurationNs'])/1e6:8.2f} ms n={r['Calls']:>5} avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:100]}")
PY
  rm -rf $O
done
