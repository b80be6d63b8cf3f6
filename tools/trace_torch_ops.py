"""Which lines of the Python mirror issue torch operations during one blurfactory training iteration?  A TorchFunctionMode counts every
torch API call by the innermost frame inside evdeblurnerf_amd/ or tools/ (autograd runs single-threaded so that the backward of the
custom Functions is seen too).  GPU box only.   python tools/trace_torch_ops.py [--awp fused]"""
import argparse
import collections
import os
import sys
import traceback
from types import SimpleNamespace

import torch
from torch.overrides import TorchFunctionMode

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_train_step as B  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Count(TorchFunctionMode):
    def __init__(self):
        super().__init__()
        self.n = collections.Counter()
        self.on = False

    def __torch_function__(self, func, types, args=(), kwargs=None):
        if self.on:
            name = getattr(func, "__name__", str(func))
            site = "?"
            for fr in reversed(traceback.extract_stack(limit=14)[:-1]):
                if fr.filename.startswith(ROOT) and "trace_torch_ops" not in fr.filename:
                    site = f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno}"
                    break
            self.n[(site, name)] += 1
        return func(*args, **(kwargs or {}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--awp", default="none")
    a = ap.parse_args()
    torch.autograd.set_multithreading_enabled(False)
    cnt = Count()
    orig = B.torch.cuda.Event

    class Ev:                                   # the timed loop of bench_train_step.run starts at its first Event.record
        def __init__(self, **kw):
            self.e = orig(**kw)

        def record(self):
            cnt.on = not cnt.on
            self.e.record()

        def synchronize(self):
            self.e.synchronize()

        def elapsed_time(self, o):
            return self.e.elapsed_time(o.e)

    B.torch.cuda.Event = Ev
    with cnt:
        B.run(SimpleNamespace(precision="f16", iters=1, pixels=1024, events=4096, P=10, awp=a.awp, mam="corr"))
    skip = {"__get__", "size", "dim", "data_ptr", "numel", "is_contiguous", "stride", "__getitem__", "reshape", "view", "detach", "shape",
            "device", "dtype", "_version", "requires_grad", "is_cuda", "grad", "__set__", "expand", "transpose", "t", "unsqueeze", "squeeze", "permute"}
    rows = [(k, v) for k, v in cnt.n.items() if k[1] not in skip]
    print("torch calls in one iteration that may launch a kernel:", sum(v for _, v in rows))
    for (site, name), v in sorted(rows, key=lambda r: -r[1])[:90]:
        print(f"{v:5d}  {name:22s} {site}")


if __name__ == "__main__":
    main()
