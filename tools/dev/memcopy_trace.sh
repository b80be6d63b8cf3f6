#!/bin/bash
# developer script: every memory copy of a few training iterations with its size and the kernels around it (rocprofv3 traces)
O=$GRAFT_REPO_ROOT/gpurun_out/r4b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/mc -- python $GRAFT_REPO_ROOT/tools/bench_train_step.py --iters 4 > $O/mc.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, os
big = lambda pat: max(glob.glob(pat), key=os.path.getsize)
d = big("gpurun_out/r4b/mc/*/*memory_copy_trace.csv")
k = big("gpurun_out/r4b/mc/*/*kernel_trace.csv")
print(d, k)
copies = list(csv.DictReader(open(d)))
kern = list(csv.DictReader(open(k)))
print("columns:", list(copies[0].keys()), len(copies), "copies", len(kern), "kernels")
ks = sorted((int(r["Start_Timestamp"]), r["Kernel_Name"][:70]) for r in kern)
import bisect
starts = [t for t, _ in ks]
t_end = max(starts)
print("kernel ts range", starts[0], t_end, "copy ts range", min(int(c["Start_Timestamp"]) for c in copies), max(int(c["Start_Timestamp"]) for c in copies))
last = copies
import collections
agg = collections.Counter(); dur = collections.Counter()
for c in last:
    t = int(c["Start_Timestamp"]); i = bisect.bisect_left(starts, t)
    prev = ks[i - 1][1] if i > 0 else "-"; nxt = ks[i][1] if i < len(ks) else "-"
    key = (c.get("Direction", "?"), c.get("Bytes", c.get("Size", "?")), prev.split("(")[0][-45:], nxt.split("(")[0][-45:])
    agg[key] += 1; dur[key] += int(c["End_Timestamp"]) - int(c["Start_Timestamp"])
for key, n in agg.most_common(40):
    print(n, f"{dur[key] / n / 1e3:8.1f} us", key)
PY
rm -rf gpurun_out/r4b/mc
