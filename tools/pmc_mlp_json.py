#!/usr/bin/env python
"""Turn the counter CSVs of tools/pmc_mlp.sh into the JSON bench.py reads (profiles/rNN_pmc_mlp.json).
    python tools/pmc_mlp_json.py gpurun_out/pmc_r02 profiles/r02_pmc_mlp.json"""
import collections, csv, glob, json, sys

root, out = sys.argv[1], sys.argv[2]
NAMES = {"k_nerf_mlp_c<": "f16c", "k_nerf_mlp<3,": "f16", "k_nerf_mlp<2,": "bf16", "k_nerf_mlp<1,": "f16x3"}
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace(" ", "")
        for pat, prec in NAMES.items():
            if pat in k:
                vals[prec][r["Counter_Name"]].append(float(r["Counter_Value"]))
                vals[prec]["_kernel"] = k.split("(")[0].replace("voidevd::", "")
kern, der = {}, {"note": "GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); "
                         "HBM traffic = 2 x FETCH_SIZE (gfx950: wide 16-B/lane loads are tallied at half, MI355X_MICROARCH.md HBM section) + WRITE_SIZE"}
for prec, d in vals.items():
    m = {c: sum(v) / len(v) for c, v in d.items() if c != "_kernel"}
    kern[prec] = {"kernel": d["_kernel"], **{c: m[c] for c in sorted(m) if c not in ("FETCH_SIZE", "WRITE_SIZE", "GRBM_COUNT")},
                  "FETCH_SIZE_KB": m.get("FETCH_SIZE"), "WRITE_SIZE_KB": m.get("WRITE_SIZE")}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
        der[prec] = {"mfma_busy_frac": m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024),
                     "traffic_bytes": (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024 if "FETCH_SIZE" in m and "WRITE_SIZE" in m else None,
                     "effective_clock_ghz_note": "GRBM_GUI_ACTIVE / 8 / kernel duration"}
json.dump({"source": "tools/pmc_mlp.sh (rocprofv3 --kernel-trace --pmc ..., one pass per counter group) on MI355X; per dispatch at 4096 rays x 128 samples",
           "kernels": kern, "derived": der}, open(out, "w"), indent=1)
print(json.dumps(der, indent=1))
