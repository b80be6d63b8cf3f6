#!/bin/bash
# developer A/B (GPU box): the f16c render's coarse level with hardware sines behind the two-float revolution reduction (default) against the float32-grade
# polynomial (EVD_COARSE_TRIG=exact): trained-parameter parity (tools/trained_c2f.py), parity tests, render time
out=gpurun_out/r06_coarse_trig_ab.log; : > $out
python tools/trained_c2f.py --iters 3000 --save /tmp/c2f_trained.npz --modes f16c 2>&1 | grep -E "^trained|^seed" >> $out
echo "== EVD_COARSE_TRIG=exact" >> $out
EVD_COARSE_TRIG=exact python tools/trained_c2f.py --load /tmp/c2f_trained.npz --modes f16c 2>&1 | grep -E "^trained" >> $out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_c2f_trained.py -q -x 2>&1 | tail -1 >> $out
for r in 1 2 3; do for t in rev exact; do
  echo "== $t (round $r)" >> $out
  EVD_COARSE_TRIG=$t python tools/bench_c2f.py --precision f16c --iters 50 2>&1 | tail -1 >> $out
done; done
