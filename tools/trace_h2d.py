"""Which Python lines of one blurfactory training iteration move data between host and device (a synchronising pageable copy each)?
A TorchDispatchMode reports every aten op whose inputs and outputs are not all on one device, with the innermost repository frame.
GPU box only.    python tools/trace_h2d.py [--awp fused]"""
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


class Watch(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.n = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        devs = {t.device.type for t in tree_flatten((args, kwargs, out))[0] if isinstance(t, torch.Tensor)}
        if len(devs) > 1:
            site = "?"
            for fr in reversed(traceback.extract_stack()[:-1]):
                if fr.filename.startswith(ROOT) and "trace_h2d" not in fr.filename:
                    site = f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno}"
                    break
            else:
                fr = traceback.extract_stack()[-3]
                site = f"{fr.filename.split('site-packages/')[-1].split('dist-packages/')[-1]}:{fr.lineno}"
            self.n[(str(func), site)] += 1
        return out


ap = argparse.ArgumentParser()
ap.add_argument("--awp", default="none")
ap.add_argument("--precision", default="f16")
a = ap.parse_args()
ns = SimpleNamespace(precision=a.precision, iters=1, pixels=1024, events=4096, P=10, awp=a.awp, mam="corr")
B.run(ns)
w = Watch()
ns.iters = 2
with w:
    B.run(ns)
print("host <-> device ops over 5 iterations (3 warm-up + 2 timed) + the set-up of the run:")
for (f, site), c in w.n.most_common(40):
    print(f"{c:5d}  {f:40s} {site}")
