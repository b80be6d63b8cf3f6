"""Steady-state kernel breakdown of one blurfactory training iteration (tools/bench_train_step.py) by torch.profiler: the iterations are
profiled AFTER the warm-up, so one-off library tuning launches do not pollute the table as they do under rocprofv3.  GPU box only.
    python tools/profile_train_kernels.py [--awp fused]"""
import argparse
import os
import sys
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench_train_step as B  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--awp", default="none")
ap.add_argument("--precision", default="f16")
ap.add_argument("--top", type=int, default=45)
a = ap.parse_args()
ns = SimpleNamespace(precision=a.precision, iters=6, pixels=1024, events=4096, P=10, awp=a.awp, mam="corr")
from torch.profiler import profile, ProfilerActivity  # noqa: E402
B.run(ns)                          # builds, tunes, caches
ns.iters = 5
prof = profile(activities=[ProfilerActivity.CUDA])
ns.timed_ctx = prof                # profiled: the timed iterations only (not the model construction and its grid uploads, not the warm-up)
ms, _, _ = B.run(ns)
torch.cuda.synchronize()
n = ns.iters
# (key_averages also lists the runtime's launch calls, which carry no device time: round 4's header counted them as kernels)
ka = sorted((e for e in prof.key_averages() if e.device_time_total > 0), key=lambda e: -e.device_time_total)
tot = sum(e.device_time_total for e in ka)
print(f"iteration {ms:.2f} ms; device time per iteration {tot / n / 1e3:.2f} ms in {sum(e.count for e in ka) / n:.0f} kernels")
lib = [e for e in ka if "evd::" in e.key or e.key.startswith("k_")]
print(f"  library kernels (evd::): {sum(e.count for e in lib) / n:.0f} launches, {sum(e.device_time_total for e in lib) / n / 1e3:.2f} ms; everything else "
      f"{(tot - sum(e.device_time_total for e in lib)) / n / 1e3:.2f} ms (launch counts of the torch side: the rocprofv3 kernel stats, tools/final_run.sh)")
for e in ka[:a.top]:
    print(f"{e.device_time_total / n:9.1f} us  n={e.count / n:6.1f}  {e.key[:110]}")
