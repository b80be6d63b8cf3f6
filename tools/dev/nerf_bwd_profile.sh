#!/bin/bash
# developer profile (GPU box): per-kernel table of the 8 x 256 NeRF's training forward + backward (tools/bench_train.py) -> gpurun_out/r06_nerf_bwd_kernels.txt
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r06_nerf_bwd_prof; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $ROOT/tools/bench_train.py --precs f16 --iters 20 > $O/run.log 2>&1
python - <<PY
import csv, glob
rows = []
for f in glob.glob("$O/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
out = open("$ROOT/gpurun_out/r06_nerf_bwd_kernels.txt", "w")
out.write(open("$O/run.log").read()[-1500:] + "\n")
for r in rows[:30]:
    out.write(f"{float(r['TotalDurationNs']) / 1e6:9.2f} ms  n={r['Calls']:>5}  avg {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:120]}\n")
PY
rm -rf $O
