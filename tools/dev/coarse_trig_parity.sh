#!/bin/bash
# developer check (GPU box): bench.py's c2f parity block (4096-ray renders against the oracle on 256 of the rays; seed and in-run trained parameters) with the
# coarse level's hardware sines (default) and with the polynomial (EVD_COARSE_TRIG=exact)
for t in rev exact; do
  EVD_COARSE_TRIG=$t python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train --no-modes --no-composite --no-awp --no-strong --no-zero-probe > gpurun_out/r06_trig_$t.json 2> gpurun_out/r06_trig_$t.err
done
python - <<PY > gpurun_out/r06_coarse_trig_parity.log
import json
for t in ("rev", "exact"):
    d = json.loads(open("gpurun_out/r06_trig_%s.json" % t).read().strip().splitlines()[-1])["c2f"]
    p = d["parity"]["rgb_linf_vs_oracle"]
    print(t, "c2f ms", round(d["ms_per_step"], 4), {k: v.get("f16c") for k, v in p.items()})
PY
