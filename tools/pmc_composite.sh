#!/bin/bash
# HBM traffic (PMC) and duration of the compositing scan on 2^20 rays x 128 samples (GPU box).
# usage: tools/pmc_composite.sh <outdir>   -- separate passes: FETCH_SIZE, WRITE_SIZE (TCC slots), kernel trace only
OUT=${1:-gpurun_out/pmc_composite}
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $ROOT/$OUT/$c -- python $ROOT/tools/probes/composite_probe.py > $ROOT/$OUT/$c.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace -- python $ROOT/tools/probes/composite_probe.py > $ROOT/$OUT/trace.log 2>&1
python - <<PY
import csv, glob, collections
root = "$ROOT/$OUT"
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(root + "/" + c + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_composite" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items(): print(f"k_composite* {k}: median {sorted(v)[len(v)//2]:.1f} KB per dispatch, min {min(v):.1f}, max {max(v):.1f} ({len(v)} dispatches: full / no weights store / pdrf layout)")
for f in glob.glob(root + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_composite" in r["Name"]: print("duration:", r["Name"][:60], r["Calls"], "calls avg", float(r["AverageNs"]) / 1e3, "us  min", float(r["MinNs"]) / 1e3)
PY
