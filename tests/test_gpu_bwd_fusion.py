"""Round 5's fused backward kernels against the per-layer chains they replace, through the library's own switches:
k_voxel_bwd_fused64 (EVD_BWD_FUSE64), k_wgrad_dgrad<..., 5, 9> (EVD_BWD_FUSE_SG), YGEN (EVD_BWD_YGEN), k_awp_bwd_fused
(EVD_AWP_BWD_FUSE), the feature gradient written as rows by the dgrad kernels (EVD_BWD_ROWS).  The switches are read once per process, so tools/dev/bwd_fusion_ab.py runs in two subprocesses."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(path, off, small=False):
    env = dict(os.environ)
    env.pop("AB_SMALL", None)
    if small:
        env["AB_SMALL"] = "1"
    for k in ("EVD_BWD_FUSE64", "EVD_BWD_FUSE_SG", "EVD_BWD_YGEN", "EVD_AWP_BWD_FUSE", "EVD_BWD_ROWS"):
        env.pop(k, None)
        if off:
            env[k] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dev", "bwd_fusion_ab.py"), path], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("small", [False, True])
def test_fused_backward_forms_equal_the_per_layer_chains(tmp_path, small):
    """small: at most one tile per wavefront; otherwise ~2 (64-wide level) / ~5 (AWP embedding) tiles per wavefront of the persistent kernels"""
    a, b = str(tmp_path / "fused.npz"), str(tmp_path / "chain.npz")
    _run(a, off=False, small=small)
    _run(b, off=True, small=small)
    fa, fb = np.load(a), np.load(b)
    assert set(fa.files) == set(fb.files) and len(fa.files) >= 30
    worst = {}
    for k in fa.files:
        x, y = fa[k].astype(np.float64), fb[k].astype(np.float64)
        assert np.isfinite(x).all() and np.isfinite(y).all(), k
        worst[k] = float(np.linalg.norm(x - y) / max(np.linalg.norm(y), 1e-30))
    print("fused vs per-layer backward, relative L2 per tensor:", {k: f"{v:.1e}" for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:6]})
    # the same half-precision products and masks; float32 partial sums in another order, and (fused forms) ReLU patterns from the stored
    # activations instead of the bit masks (equal unless a positive pre-activation rounds to zero in half precision)
    # measured: 3.4e-7 worst (the basis gradients, accumulated by atomics), the network gradients below 1e-7
    for k, v in worst.items():
        assert v < 1e-5, (k, v)
