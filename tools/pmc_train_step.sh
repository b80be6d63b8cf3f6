#!/bin/bash
# HBM traffic + matrix-pipe counters of the kernels of one blurfactory TRAINING iteration (tools/bench_train_step.py), GPU box.
# usage: tools/pmc_train_step.sh <outdir> [precision]    Separate --pmc passes (FETCH_SIZE / WRITE_SIZE do not share a pass); kernel trace only.
OUT=${1:-gpurun_out/pmc_train_step}
PREC=${2:-f16}
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "GRBM_GUI_ACTIVE"; do
  d=$(echo $c | cut -d' ' -f1)
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $ROOT/$OUT/$d -- python $ROOT/tools/bench_train_step.py --precision $PREC --iters 3 > $ROOT/$OUT/$d.log 2>&1
done
python - <<PY > $ROOT/$OUT/summary.txt
import csv, glob, collections
root = "$ROOT/$OUT"
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:64]
        if "evd::" not in k: continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
print("per-dispatch averages over the run's dispatches (iterations + warm-up; batch sizes differ between the blur and the event batches).")
print("FETCH_SIZE / WRITE_SIZE in rocprofv3 KB; FETCH_SIZE x 2 = bytes read for 16-byte-per-lane streaming (MI355X_MICROARCH.md, HBM section);")
print("mfma busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)")
for k in sorted(tot, key=lambda k: -tot[k].get("FETCH_SIZE", 0)):
    a = lambda c: tot[k].get(c, 0) / max(n[(k, c)], 1)
    f, w, mf, gui = a("FETCH_SIZE"), a("WRITE_SIZE"), a("SQ_VALU_MFMA_BUSY_CYCLES"), a("GRBM_GUI_ACTIVE")
    busy = mf / (gui / 8 * 1024) if gui else 0
    print(f"{k:66s} read {2 * f / 1024:9.1f} MB  written {w / 1024:9.1f} MB  mfma busy {100 * busy:5.1f} %  VALU insts {a('SQ_INSTS_VALU'):12.0f}  dispatches {n[(k, 'FETCH_SIZE')]}")
PY
head -40 $ROOT/$OUT/summary.txt
