"""Builds libevdnerf.so (HIP, gfx950 only) in-tree with hipcc.

    python -m evdeblurnerf_amd.build [--force]

Every csrc/*.hip is compiled to an object next to the library and linked into
evdeblurnerf_amd/lib/libevdnerf.so; the .so travels to the GPU box with the
repo snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libevdnerf.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function", "-Wno-unused-variable"]
# Per-file flags.  The one-wavefront-per-SIMD MLP kernels (f16c: 494 registers, f16x3: 420) keep part of their state in AGPRs; by default
# hipcc puts the MFMA accumulators there, and every epilogue value then costs a v_accvgpr_read (plus a write-back for the values that stay).
# With the MFMAs in VGPR form the accumulators live in VGPRs and the MFMA-only operands (fp6 blocks, weight fragments) go to the AGPRs, which
# the matrix core reads directly: f16c 5708 -> 4802 VALU instructions per wavefront, 0.879 -> 0.857 ms; f16x3 1.58 -> 1.53 ms
# (tools/run_lib_variants.sh; the two-wavefront kernels and the training / scatter kernels do not change or get slower with it).
VGPR_FORM = ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]
# The TRAIN variants of the f16c kernels add ~70 stores and the fragment permutations to a body that hipcc unrolls completely (#pragma unroll
# over the static layer table): past the default pragma-unroll threshold it leaves the loops rolled, indexes the register arrays dynamically
# and the kernel runs out of scratch memory (1900 branches, 4000 scratch accesses).  The inference units are compiled without the flag.
UNROLL = ["-mllvm", "-pragma-unroll-threshold=1000000"]
PER_FILE = {"kernel_nerf_mlp_pipe_f16c.hip": VGPR_FORM, "kernel_nerf_mlp_pipe_f16x3.hip": VGPR_FORM, "kernel_voxel_pipe_f16c.hip": VGPR_FORM,
            "kernel_voxel_train_f16c.hip": VGPR_FORM + UNROLL + ["-DEVD_C_RNE"], "kernel_nerf_train_fwd_f16c.hip": VGPR_FORM + UNROLL + ["-DEVD_C_RNE"]}
# -DEVD_C_RNE on the TRAIN units: the float16 part of an activation rounded to nearest instead of truncated (mlp_pipe_c.h c_drain_pair; the
# inference kernels truncate: one VALU instruction less per pair).  The fp6 residual is then half as large, the pre-activations twice as
# accurate -- half as many ReLU units decided differently from float32: gradient error (median over the 31 tensors of
# tools/train_parity.py) 3.6e-3 -> 2.5e-3 of the norm at the same iteration time (the training forward is 10 % of the iteration).


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libevdnerf.so cannot be built")
    return exe


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "evdnerf.h"))
    hs.append(os.path.abspath(__file__))          # the flags live here
    return hs


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, obj, extra):
    cmd = [hipcc(), *FLAGS, *PER_FILE.get(os.path.basename(src), []), *extra, "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {os.path.basename(src)}:\n{r.stdout}\n{r.stderr}")
    return r.stderr


def build(force: bool = False, verbose: bool = False, extra=()) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    hs = headers()
    jobs = []
    objs = []
    for src in sources():
        obj = os.path.join(LIBDIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hs):
            jobs.append((src, obj))
    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for (src, _), warn in zip(jobs, ex.map(lambda j: _compile(j[0], j[1], list(extra)), jobs)):
                if verbose and warn.strip():
                    print(f"[{os.path.basename(src)}]\n{warn}", file=sys.stderr)
    if jobs or _stale(LIB, objs):
        cmd = [hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
