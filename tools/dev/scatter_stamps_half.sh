out=gpurun_out/r06_scatter_stamps_half.log; : > $out
for v in vbwstamp vbwstamp_na; do for p in f32 f16; do
  echo "== $v, re-gather grids: $p" >> $out
  EVD_STAMP_PREC=$p EVD_LIB_PATH=$PWD/evdeblurnerf_amd/lib/variants/libevd_$v.so python tools/dev/stamp_scatter_w.py 2>&1 | grep -v "Warn\|amdgpu.ids" >> $out
done; done
