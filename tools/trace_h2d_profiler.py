"""Where do the host-to-device copies of one blurfactory training iteration come from?  torch.profiler (CPU + GPU activities, Python
stacks): every GPU 'Memcpy HtoD' event is matched through its correlation id to the runtime call that issued it, and that call to the
innermost enclosing aten operator / Python frame.  GPU box only.    python tools/trace_h2d_profiler.py [--precision f16] [--awp fused]"""
import argparse
import collections
import json
import os
import sys
import tempfile
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench_train_step as B  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--awp", default="none")
ap.add_argument("--precision", default="f16")
a = ap.parse_args()
ns = SimpleNamespace(precision=a.precision, iters=3, pixels=1024, events=4096, P=10, awp=a.awp, mam="corr")
B.run(ns)
ns.iters = 4
prof = profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True)
ns.timed_ctx = prof                # the timed iterations only (bench_train_step.run)
B.run(ns)
torch.cuda.synchronize()
path = os.path.join(tempfile.gettempdir(), "h2d_trace.json")
prof.export_chrome_trace(path)
ev = json.load(open(path))["traceEvents"]
gpu = [e for e in ev if e.get("cat") == "gpu_memcpy" and "HtoD" in e.get("name", "")]
rt = {e["args"]["correlation"]: e for e in ev if e.get("cat") in ("cuda_runtime", "cuda_driver") and "correlation" in e.get("args", {})}
cpu = [e for e in ev if e.get("cat") in ("cpu_op", "python_function", "user_annotation") and "dur" in e]
by_tid = collections.defaultdict(list)
for e in cpu:
    by_tid[e["tid"]].append(e)
agg, byt = collections.Counter(), collections.Counter()
n_iter = ns.iters
for g in gpu:
    r = rt.get(g["args"].get("correlation"))
    if r is None:
        agg[("?", "?")] += 1
        continue
    enc = [e for e in by_tid[r["tid"]] if e["ts"] <= r["ts"] and e["ts"] + e["dur"] >= r["ts"] + r.get("dur", 0)]
    ops = sorted((e for e in enc if e["cat"] == "cpu_op"), key=lambda e: e["dur"])
    py = sorted((e for e in enc if e["cat"] == "python_function" and ("/repo/" in e["name"] or "evdeblurnerf" in e["name"] or "tools/" in e["name"])), key=lambda e: e["dur"])
    key = (r["name"], ops[0]["name"] if ops else "-", py[0]["name"][-90:] if py else "-")
    agg[key] += 1
    byt[key] += g["args"].get("bytes", 0)
print(f"{len(gpu)} 'Memcpy HtoD' GPU events over {n_iter} iterations = {len(gpu) / n_iter:.1f} per iteration; by issuing call / aten op / innermost repository frame:")
for key, c in agg.most_common(30):
    print(f"{c / n_iter:6.1f} per iteration  {byt[key] / max(c, 1):10.0f} B  {key}")
