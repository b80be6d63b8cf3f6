#!/usr/bin/env python
"""Build ablation variants of the fused MLP kernel (compile-time -DEVD_* switches in csrc/mlp_pipe.h / mlp_device.h;
EVD_ABLATE_FILE selects the translation unit, default the bf16 pipelined kernel) next to the
product library.  Ablated variants compute WRONG results on purpose; they exist to price one part of the kernel.

    python tools/ablate.py build VARIANT[=VALUE] ...     (here, cross-compiles)
    python tools/ablate.py run   VARIANT ...             (GPU box: tools/bench_mlp.py per variant)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from evdeblurnerf_amd import build as B  # noqa: E402


def lib_of(v):
    return os.path.join(B.LIBDIR, "abl_" + v.replace("=", "_") + ".so")


def build(variants):
    B.build()
    target = os.environ.get("EVD_ABLATE_FILE", "kernel_nerf_mlp_pipe_bf16.hip")
    others = [os.path.join(B.LIBDIR, os.path.basename(s)[:-4] + ".o") for s in B.sources() if not s.endswith(target)]
    src = os.path.join(B.CSRC, target)

    def one(v):
        obj = os.path.join(B.LIBDIR, "abl_" + v.replace("=", "_") + ".o")
        defs = [] if v == "baseline" else ["-DEVD_" + d for d in v.split("+")]
        subprocess.check_call([B.hipcc(), *B.FLAGS, *defs, "-c", src, "-o", obj])
        subprocess.check_call([B.hipcc(), "-shared", "-fPIC", f"--offload-arch={B.ARCH}", obj, *others, "-o", lib_of(v)])
        os.remove(obj)
        print("built", lib_of(v), flush=True)

    import concurrent.futures as cf
    with cf.ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(one, variants))


def run(variants, extra):
    for v in variants:
        print("==", v, flush=True)
        env = dict(os.environ, EVD_LIB_PATH=lib_of(v))
        script = os.environ.get("EVD_ABLATE_BENCH", "bench_mlp.py")      # e.g. bench_c2f.py
        subprocess.call([sys.executable, os.path.join(ROOT, "tools", script), *extra], env=env)


if __name__ == "__main__":
    cmd, rest = sys.argv[1], sys.argv[2:]
    vs = [r for r in rest if not r.startswith("--")]
    ex = [r for r in rest if r.startswith("--")]
    build(vs) if cmd == "build" else run(vs, ex)
