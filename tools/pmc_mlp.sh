#!/bin/bash
# PMC counters of the fused MLP kernel (GPU box).  usage: tools/pmc_mlp.sh <precs> <outdir>
# Separate passes per counter group (8 SQ slots / 4 TCC slots per pass); no tracing domains besides kernel-trace.
PRECS=${1:-bf16}
OUT=${2:-gpurun_out/pmc}
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
run() {  # name, counters
  rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $ROOT/$OUT/$1 -- python $ROOT/tools/bench_mlp.py --precs=$PRECS --iters 5 > $ROOT/$OUT/$1.log 2>&1
}
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"
run sq2 "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"
run grbm "GRBM_GUI_ACTIVE GRBM_COUNT"
run tcc1 "FETCH_SIZE"
run tcc2 "WRITE_SIZE"
python - <<PY
import csv, glob, collections, os
root = "$ROOT/$OUT"
for d in sorted(glob.glob(root + "/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        disp = collections.Counter()
        for r in csv.DictReader(open(f)):
            disp[(r["Kernel_Name"][:60], r["Counter_Name"])] += 1
        for k, v in agg.items():
            if "mlp" not in k: continue
            print(os.path.basename(os.path.dirname(d.rstrip("/"))) if False else d.split("/")[-2], k)
            for c, x in sorted(v.items()):
                n = disp[(k, c)]
                print(f"    {c:32s} {x / n:16.1f} per dispatch ({n} dispatches)")
PY
