#!/bin/bash
# PMC counters of k_voxel_mlp_c (PDRF fine level, compensated float16) inside a c2f render (GPU box).  usage: tools/pmc_voxel_c.sh <outdir>
# Separate passes per counter group; no tracing domains besides kernel-trace.
OUT=${1:-gpurun_out/pmc_voxel_c}
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
run() {  # name, counters
  rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $ROOT/$OUT/$1 -- python $ROOT/tools/bench_c2f.py --precision f16c --iters 5 > $ROOT/$OUT/$1.log 2>&1
}
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"
run sq2 "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD"
run grbm "GRBM_GUI_ACTIVE GRBM_COUNT"
run tcc1 "FETCH_SIZE"
run tcc2 "WRITE_SIZE"
python - <<PY
import csv, glob, collections
root = "$ROOT/$OUT"
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_voxel_mlp_c" in k or "k_nerf_mlp_c" in k:
            tot[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in tot.items():
    print(k)
    m = {c: sum(x) / len(x) for c, x in v.items()}
    for c in sorted(m):
        print(f"    {c:32s} {m[c]:16.1f} per dispatch ({len(v[c])} dispatches)")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
        print(f"    matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) = {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (m['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f}")
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        print(f"    HBM traffic = 2 x FETCH_SIZE + WRITE_SIZE = {(2 * m['FETCH_SIZE'] + m['WRITE_SIZE']) / 1024:.2f} MB per dispatch")
PY
rm -rf $ROOT/$OUT/*/
