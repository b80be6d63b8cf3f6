"""The f16c kernels drive their LDS-DMA (global_load_lds: the LDS base is M0) through inline asm that sets M0 ONCE per 16 KiB chunk and relies on
nothing else touching it between the pieces (csrc/mlp_pipe_c.h CStream::issue_piece).  The compiler cannot see that dependency, so this test
disassembles the built kernels and fails on any M0 writer other than `s_mov_b32 m0, sN` and on any instruction family that uses M0 implicitly
for something else (relative register addressing, s_set_gpr_idx, GDS / sendmsg) -- e.g. a register array that a failed unroll left dynamically
indexed (seen while building the TRAIN variants: build.py UNROLL).  CPU-only: works on the objects build() leaves in evdeblurnerf_amd/lib."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "evdeblurnerf_amd", "lib")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
UNITS = ["kernel_nerf_mlp_pipe_f16c.o", "kernel_voxel_pipe_f16c.o", "kernel_voxel_train_f16c.o", "kernel_nerf_train_fwd_f16c.o"]


@pytest.mark.parametrize("unit", UNITS)
def test_only_the_dma_asm_writes_m0(unit, tmp_path):
    obj = os.path.join(LIBDIR, unit)
    if not os.path.exists(OBJDUMP) or not os.path.exists(obj):
        pytest.skip("llvm-objdump or the built object is missing (run __graft_entry__.build())")
    local = tmp_path / unit
    shutil.copy(obj, local)
    subprocess.run([OBJDUMP, "--offloading", str(local)], check=True, capture_output=True)
    dev = [f for f in os.listdir(tmp_path) if "amdgcn" in f]
    assert dev, "no device code object in " + unit
    dis = subprocess.run([OBJDUMP, "-d", str(tmp_path / dev[0])], check=True, capture_output=True, text=True).stdout
    kernels = re.split(r"\n(?=[0-9a-f]{16} <)", dis)
    seen = 0
    for k in kernels:
        head = k.split("\n", 1)[0]
        if "_mlp_c" not in head:           # k_nerf_mlp_c / k_voxel_mlp_c instantiations
            continue
        seen += 1
        writes = 0
        for line in k.split("\n")[1:]:
            ins = line.split("//")[0].strip()
            if not ins:
                continue
            m = re.match(r"(\S+)\s*(.*)", ins)
            op, args = m.group(1), m.group(2)
            assert not re.search(r"movrel|s_set_gpr_idx|s_sendmsg\b|_gds\b|ds_gws", op), f"{head}: {ins} uses M0 implicitly"
            dst = args.split(",")[0].strip()
            if dst == "m0":
                assert op == "s_mov_b32" and re.match(r"m0, s\d+$", args), f"{head}: unexpected M0 writer: {ins}"
                writes += 1
        assert writes > 0, head + ": no M0 set-up found (not a DMA kernel?)"
    assert seen > 0, "no f16c kernel in " + unit
