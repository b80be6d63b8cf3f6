"""`bench.py --gpus N` is its own launcher (the flag used to be parsed and ignored).  CPU checks of that path:
the decision and the command line, and a real 2-rank launch of bench.py's rank scaffolding under gloo with a stub step."""
import json
import os
import subprocess
import sys

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_launch_decision_and_command():
    a = bench.parse(["--gpus", "4", "--steps", "7"])
    assert bench.needs_launch(a, env={})                                  # no launcher environment: bench.py starts the ranks
    assert not bench.needs_launch(a, env={"WORLD_SIZE": "4", "RANK": "0"})  # under torchrun (the driver's form): join as a rank
    assert not bench.needs_launch(bench.parse([]), env={})                # N = 1: plain single process
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "7"], port=29511)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-5] == os.path.join(ROOT, "bench.py") and cmd[-4:] == ["--gpus", "4", "--steps", "7"]


def test_two_rank_launch_under_gloo():
    """the launch path end to end: torch.distributed.run starts 2 ranks of the stub, they rendezvous on 127.0.0.1, run
    warm-up + exactly K timed steps with an all-reduce per step, and rank 0 prints ONE JSON line with n_gpus = 2 whose time is
    the slowest rank's"""
    argv = ["--gpus", "2", "--steps", "6", "--warmup", "2", "--rays", "4096"]
    cmd = bench.launch_command(2, argv, script=os.path.join(ROOT, "tests", "_bench_rank_stub.py"))
    env = dict(os.environ, EVD_BENCH_SELF_LAUNCH="1", OMP_NUM_THREADS="1", CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 6 and j["warmup"] == 2 and j["scaling"] == "weak"
    assert j["ranks"] == {"world_size": 2, "backend": "gloo", "self_launch": True, "reduced": 3.0}
    assert j["ms_per_step"] >= 4.0                                        # rank 1 sleeps 4 ms per step: MAX over ranks
    assert abs(j["value"] - 2 * 4096 * 6 / (j["ms_per_step"] * 6e-3)) < 1e-6 * j["value"]
