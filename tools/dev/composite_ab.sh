#!/bin/bash
# A/B of the compositing scan's forms on one box: tools/dev/composite_ab.sh <out.log>
O=${1:-gpurun_out/composite_ab.log}
P=tools/probes/composite_probe.py
{
[ -x tools/probes/bin/hbm_mix_probe ] && tools/probes/bin/hbm_mix_probe
PROBE_YARDSTICKS=1 EVD_COMPOSITE_FORM=rows python $P
for r in 1 2 4; do for nt in 0 1; do EVD_COMPOSITE_FORM=il EVD_COMPOSITE_RPW=$r EVD_COMPOSITE_NT=$nt python $P | sed "s/^\[/[rpw=$r nt=$nt /"; done; done
EVD_COMPOSITE_FORM=rows python $P
for S in 64 192 256; do for f in rows il; do PROBE_S=$S EVD_COMPOSITE_FORM=$f EVD_COMPOSITE_NT=1 python $P | sed "s/^\[/[S=$S /"; done; done
} 2>&1 | grep -v "amdgpu.ids\|Warn" | tee $O
