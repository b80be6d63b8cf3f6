#!/bin/bash
# developer A/B (GPU box): the 64-wide level's inference pass, resident stream (default; variant res256 = a build with -DEVD_RES_NT=256: one wavefront per SIMD) against the streaming kernel
out=gpurun_out/r06_coarse_form_ab.log; : > $out
for p in f16x3 f16; do
  python tools/dev/coarse_form_check.py /tmp/res_$p.npy $p 2>&1 | grep "level forward" >> $out
  EVD_COARSE_FORM=pipe python tools/dev/coarse_form_check.py /tmp/pipe_$p.npy $p 2>&1 | grep "level forward" >> $out
  python -c "import numpy as np; a=np.load('/tmp/res_$p.npy'); b=np.load('/tmp/pipe_$p.npy'); print('$p: resident vs pipe: max |diff| =', float(np.abs(a-b).max()), 'over', a.size, 'values; bit-equal:', bool((a.view(np.uint32)==b.view(np.uint32)).all()))" >> $out
done
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_c2f_trained.py -q -x 2>&1 | tail -1 >> $out
for r in 1 2 3; do for form in resident res256 pipe; do for p in f16c f16; do
  unset EVD_LIB_PATH EVD_COARSE_FORM
  [ $form = pipe ] && export EVD_COARSE_FORM=pipe
  [ $form = res256 ] && export EVD_LIB_PATH=$PWD/evdeblurnerf_amd/lib/variants/libevd_res256.so
  echo "== $form $p (round $r)" >> $out
  python tools/bench_c2f.py --precision $p --iters 50 2>&1 | tail -1 >> $out
done; done; done
