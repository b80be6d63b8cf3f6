import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import subprocess
# run the c2f bench in-process (3 gathers per render; the last launch of the run is the fine level: 16384 blocks)
sys.argv = ["bench_c2f.py", "--precision", "f16", "--iters", "3"]
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench_c2f
bench_c2f.main()
torch.cuda.synchronize()
from evdeblurnerf_amd import _lib as L
buf = np.zeros(8 * 8192, np.int64)
rc = L.lib().evd_debug_vs_trace(buf.ctypes.data_as(C.POINTER(C.c_longlong)))
t = buf.reshape(8192, 8)
d = np.diff(t, axis=1)
print("rc", rc, "stamps: 0 start | 1 after geometry+sync | 2 gather done | 3 after sync | 4 mfma done | 5 reduce done | 6 after sync | 7 stores issued")
print("median cycles between stamps (counter ticks):", np.median(d, axis=0))
print("p90:", np.percentile(d, 90, axis=0))
print("median block lifetime:", np.median(t[:, 7] - t[:, 0]), " start spread (last start - first start):", t[:, 0].max() - t[:, 0].min())
