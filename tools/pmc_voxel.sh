#!/bin/bash
# PMC counters of the tri-plane gather k_voxel_sample at the blurfactory grid sizes (GPU box): memory-side bytes, L2 hit / miss, duration.
# usage: tools/pmc_voxel.sh <outdir> [precision]   -- separate passes per counter group; no tracing domains besides kernel-trace
OUT=${1:-gpurun_out/pmc_voxel}
PREC=${2:-f16}
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
run() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $ROOT/$OUT/$1 -- python $ROOT/tools/bench_c2f.py --precision $PREC --iters 5 > $ROOT/$OUT/$1.log 2>&1; }
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
run tcp "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"
run sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAVES"
run grbm "GRBM_GUI_ACTIVE"
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace -- python $ROOT/tools/bench_c2f.py --precision $PREC --iters 5 > $ROOT/$OUT/trace.log 2>&1
python - <<PY
import csv, glob, collections
root = "$ROOT/$OUT"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_voxel_sample" in r["Kernel_Name"] and "bwd" not in r["Kernel_Name"]:
            # three launches per render: coarse (64 samples), coarse at the new points, fine (128 samples): key by grid size
            agg[r["Grid_Size"] if "Grid_Size" in r else "?"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for gsz, d in sorted(agg.items()):
    print("k_voxel_sample  grid", gsz)
    for c, v in sorted(d.items()):
        print(f"    {c:34s} {sorted(v)[len(v) // 2]:16.1f}  (median of {len(v)} dispatches)")
for f in glob.glob(root + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "voxel" in r["Name"]: print("duration:", r["Name"][:60], r["Calls"], "calls avg", float(r["AverageNs"]) / 1e3, "us  min", float(r["MinNs"]) / 1e3, "max", float(r["MaxNs"]) / 1e3)
PY
