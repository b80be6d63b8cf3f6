"""Rank body for tests/test_bench_launcher.py: bench.py's own rank scaffolding (init_ranks -> time_steps -> headline, rank 0
prints the JSON line) with the render step replaced by a CPU stand-in, so that the `--gpus N` launch path runs under gloo."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

a = bench.parse(sys.argv[1:])
rank, world, local, dist, backend, device = bench.init_ranks()
assert world == a.gpus, (world, a.gpus)
pending = []
calls = [0]
last = [None]


def step():
    calls[0] += 1
    time.sleep(0.002 * (rank + 1))                  # rank 1 is the slow one: the line must report the MAX over ranks
    p = torch.full((8,), float(rank + 1))
    if pending:
        pending.pop().wait()
    pending.append(dist.all_reduce(p, async_op=True))
    last[0] = p


dt = bench.time_steps(step, a.steps, a.warmup, dist, drain=lambda: [w.wait() for w in pending], device=device)
assert calls[0] == a.steps + a.warmup
line = bench.headline(a.rays, a.samples, world, a.steps, a.warmup, dt, a.precision)
line["ranks"] = {"world_size": world, "backend": backend, "self_launch": os.environ.get("EVD_BENCH_SELF_LAUNCH") == "1",
                 "reduced": float(last[0][0])}
if rank == 0:
    print(json.dumps(line), flush=True)
dist.barrier()
dist.destroy_process_group()
