#!/bin/bash
# PMC counters of the tri-plane scatter k_voxel_sample_bwd at the blurfactory fine-level size (GPU box): atomic requests at the L2,
# their sectors, what reaches the memory side, duration.  Separate passes per counter group; kernel-trace only.
# usage: tools/pmc_scatter.sh <outdir>
OUT=${1:-gpurun_out/pmc_scatter}
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
run() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $ROOT/$OUT/$1 -- python $ROOT/tools/bench_voxel_bwd.py --iters 3 > $ROOT/$OUT/$1.log 2>&1; }
run atomic "TCC_ATOMIC_sum TCC_ATOMIC_SECTORS_sum TCC_ATOMIC_WITHOUT_RET_REQ_sum TCC_REQ_sum"
run ea "TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum"
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run grbm "GRBM_GUI_ACTIVE"
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace -- python $ROOT/tools/bench_voxel_bwd.py --iters 3 > $ROOT/$OUT/trace.log 2>&1
python - <<PY
import csv, glob, collections
root = "$ROOT/$OUT"
agg = collections.defaultdict(list)
for f in glob.glob(root + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_voxel_sample_bwd<false" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("k_voxel_sample_bwd<false, true>, fine level 586 x 586 x 390, 524 288 samples (302 M float atomics issued), per launch:")
for c, v in sorted(agg.items()):
    print(f"    {c:36s} {sorted(v)[len(v) // 2]:18.1f}  (median of {len(v)} dispatches)")
for f in glob.glob(root + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_voxel_sample_bwd<false" in r["Name"]: print("duration:", r["Name"][:50], r["Calls"], "calls avg", float(r["AverageNs"]) / 1e3, "us  min", float(r["MinNs"]) / 1e3)
PY
