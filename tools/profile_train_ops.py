"""Which torch ops (fills, adds, copies) does one blurfactory training iteration launch besides the library's kernels?
Groups the aten ops of tools/bench_train_step.py's iteration by input shape and by Python call site.  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
import bench_train_step as B


def main():
    a = SimpleNamespace(precision="f16", iters=2, pixels=1024, events=4096, P=10, awp="none")
    from torch.profiler import profile, ProfilerActivity
    B.run(a)                                   # warm: builds, caches
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        B.run(a)
    ka = prof.key_averages(group_by_input_shape=True)
    rows = [(e.key, e.count, e.device_time_total, str(e.input_shapes)[:120]) for e in ka
            if e.key in ("aten::fill_", "aten::zero_", "aten::add_", "aten::add", "aten::copy_", "aten::zeros", "aten::mul", "aten::div", "aten::sum", "aten::cat")]
    rows.sort(key=lambda r: -r[1])
    for r in rows[:70]:
        print("%-14s n=%5d dev_us=%9.0f %s" % r)
    print("---- by stack")
    ks = prof.key_averages(group_by_stack_n=6)
    srows = [(e.key, e.count, e.device_time_total, [s for s in e.stack if "evdeblurnerf_amd" in s or "tools/" in s][:3]) for e in ks
             if e.key in ("aten::fill_", "aten::zero_", "aten::add_", "aten::add", "aten::copy_")]
    srows.sort(key=lambda r: -r[1])
    for r in srows[:60]:
        print("%-12s n=%5d dev_us=%8.0f %s" % r)


if __name__ == "__main__":
    main()
