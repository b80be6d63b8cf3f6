#!/bin/bash
# PMC counters of the tri-plane scatter k_voxel_sample_bwd at the blurfactory fine-level size (GPU box): atomic requests at the L2,
# their sectors, what reaches the memory side, duration.  Separate passes per counter group; kernel-trace only.
# usage: tools/pmc_scatter.sh <outdir> [slope]      (round 3: the wavefront-autonomous form k_voxel_sample_bwd_w behind the hybrid entry as well)
OUT=${1:-gpurun_out/pmc_scatter}
SLOPE=${2:-0.05}
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
run() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $ROOT/$OUT/$1 -- python $ROOT/tools/bench_voxel_bwd.py --iters 3 --slope $SLOPE > $ROOT/$OUT/$1.log 2>&1; }
run atomic "TCC_ATOMIC_sum TCC_ATOMIC_SECTORS_sum TCC_ATOMIC_WITHOUT_RET_REQ_sum TCC_REQ_sum"
run ea "TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum"
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run grbm "GRBM_GUI_ACTIVE"
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace -- python $ROOT/tools/bench_voxel_bwd.py --iters 3 --slope $SLOPE > $ROOT/$OUT/trace.log 2>&1
python - <<PY
import csv, glob, collections
root = "$ROOT/$OUT"
agg = collections.defaultdict(list)
for f in glob.glob(root + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for key in ("k_voxel_sample_bwd<0", "k_voxel_sample_bwd_w", "k_scatter_lines", "k_basis_grad"):
            if key in r["Kernel_Name"].replace(" ", ""):
                agg[(key, r["Counter_Name"])].append(float(r["Counter_Value"]))
print("fine level 586 x 586 x 390, 524 288 samples, slope $SLOPE, per launch (k_voxel_sample_bwd<0 = all 302 M taps by float atomics):")
for (k, c), v in sorted(agg.items()):
    print(f"    {k:24s} {c:36s} {sorted(v)[len(v) // 2]:18.1f}  (median of {len(v)} dispatches)")
for f in glob.glob(root + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("k_voxel_sample_bwd", "k_scatter_lines", "k_basis_grad")): print("duration:", r["Name"][:60], r["Calls"], "calls avg", float(r["AverageNs"]) / 1e3, "us  min", float(r["MinNs"]) / 1e3)
PY
