#!/bin/bash
# HBM traffic counters of the NeRF-mode training kernels (GPU box).  usage: tools/pmc_train.sh <outdir>
# Separate --pmc passes (FETCH_SIZE / WRITE_SIZE do not fit one pass); no tracing domains besides kernel-trace.
OUT=${1:-gpurun_out/pmc_train}
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $ROOT/$OUT/$c -- python $ROOT/tools/bench_train.py --precs f16 --iters 3 > $ROOT/$OUT/$c.log 2>&1
done
python - <<PY > $ROOT/$OUT/summary.txt
import csv, glob, collections
root = "$ROOT/$OUT"
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:70]
        if "evd::" not in k: continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
print("per-dispatch averages, rocprofv3 units (KB): FETCH_SIZE x2 = bytes read for 16-byte-per-lane streaming (MI355X_MICROARCH.md, HBM section)")
for k in sorted(tot):
    f = tot[k].get("FETCH_SIZE", 0) / max(n[(k, "FETCH_SIZE")], 1); w = tot[k].get("WRITE_SIZE", 0) / max(n[(k, "WRITE_SIZE")], 1)
    print(f"{k:72s} FETCH_SIZE {f:12.1f}  (x2 = {2 * f / 1024:9.1f} MB)   WRITE_SIZE {w:12.1f} ({w / 1024:9.1f} MB)   dispatches {n[(k, 'FETCH_SIZE')]}")
PY
cat $ROOT/$OUT/summary.txt
