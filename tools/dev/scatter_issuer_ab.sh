cd /tmp && export TMPDIR=/tmp; cd - > /dev/null
out=gpurun_out/r06_scatter_stamps.log; : > $out
export EVD_SCATTER_ISSUER=0
for v in vbwstamp vbwstamp_na; do
  echo "==== $v (round-5 form, every wavefront adds its own taps)" >> $out
  EVD_LIB_PATH=$PWD/evdeblurnerf_amd/lib/variants/libevd_$v.so python tools/dev/stamp_scatter_w.py 2>&1 | grep -v amdgpu.ids >> $out
done
unset EVD_SCATTER_ISSUER
out2=gpurun_out/r06_scatter_issuer_ab.log; : > $out2
for r in 1 2; do for iss in 0 1; do for slope in 0.05 0.35; do
  echo "== EVD_SCATTER_ISSUER=$iss slope $slope" >> $out2
  EVD_SCATTER_ISSUER=$iss python tools/bench_voxel_bwd.py --slope $slope --iters 20 2>&1 | grep "scatter backward" | sed 's/.*| scatter backward/scatter backward/' >> $out2
  EVD_SCATTER_ISSUER=$iss python tools/bench_voxel_bwd.py --slope $slope --iters 20 --dpts 2>&1 | grep "scatter backward" | sed 's/.*| scatter backward/scatter backward (d pts)/' >> $out2
done; done; done
