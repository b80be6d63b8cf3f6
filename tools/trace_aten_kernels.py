"""Which lines issue the torch-side DEVICE work of one blurfactory training iteration?  A TorchDispatchMode counts every aten op that
touches a device tensor and is not a view / allocation, by the innermost frame inside evdeblurnerf_amd/ or tools/ ('autograd' where the
op comes from the engine itself: gradient accumulation, the backward of a torch op).  GPU box only.
    python tools/trace_aten_kernels.py [--awp fused]"""
import argparse
import collections
import os
import sys
import traceback
from types import SimpleNamespace

import torch
from torch.utils._python_dispatch import TorchDispatchMode
from torch.utils._pytree import tree_flatten

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_train_step as B  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NO_KERNEL = ("view", "reshape", "empty", "as_strided", "slice", "select", "expand", "unsqueeze", "squeeze", "transpose", "permute", "detach", "alias",
             "t.default", "split", "unbind", "narrow", "_unsafe_view", "set_", "resize_", "is_", "size", "stride", "numel", "sym_", "lift_fresh",
             "_local_scalar_dense", "unfold", "chunk", "record_stream", "_has_", "item", "_to_copy" "contiguous")


class Watch(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.n = collections.Counter()
        self.on = False

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        if self.on:
            name = str(func).replace("aten.", "")
            ts = [t for t in tree_flatten((args, kwargs, out))[0] if isinstance(t, torch.Tensor)]
            if any(t.is_cuda for t in ts) and not any(k in name for k in NO_KERNEL):
                site = "autograd"
                for fr in reversed(traceback.extract_stack(limit=24)[:-1]):
                    if fr.filename.startswith(ROOT) and "trace_aten_kernels" not in fr.filename:
                        site = f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno}"
                        break
                shp = "x".join(str(d) for d in (out.shape if isinstance(out, torch.Tensor) else ts[0].shape))
                self.n[(site, name, shp)] += 1
        return out


ap = argparse.ArgumentParser()
ap.add_argument("--awp", default="none")
a = ap.parse_args()
torch.autograd.set_multithreading_enabled(False)
ns = SimpleNamespace(precision="f16", iters=1, pixels=1024, events=4096, P=10, awp=a.awp, mam="corr")
w = Watch()
orig = torch.cuda.Event


class Ev:                       # the timed loop of bench_train_step.run starts at its first Event.record
    def __init__(self, **kw):
        self.e = orig(**kw)

    def record(self):
        w.on = not w.on
        self.e.record()

    def synchronize(self):
        self.e.synchronize()

    def elapsed_time(self, o):
        return self.e.elapsed_time(o.e)


B.torch.cuda.Event = Ev
ns.iters = 2
with w:
    B.run(ns)
tot = sum(w.n.values())
print(f"aten ops with device work over 2 timed iterations: {tot} ({tot / 2:.1f} per iteration)")
for (site, f, shp), c in w.n.most_common(80):
    print(f"{c / 2:6.1f}  {f:28s} {shp:18s} {site}")
