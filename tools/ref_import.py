"""Import the read-only Python reference on CPU, in THIS container only.

Used by tools/gen_golden.py and tools/time_reference_cpu.py. Nothing under
tests/, bench.py or the package imports this module: /root/reference does
not exist on the GPU box.

The reference hard-codes CUDA and a few third-party modules that are absent
here. We stub exactly those and patch ``.cuda()`` to identity so
that its own code runs unchanged on the CPU backend of torch.
"""
from __future__ import annotations

import argparse
import os
import sys
import types

REF_ROOT = os.environ.get("EVD_REFERENCE_ROOT", "/root/reference")


def _stub(name: str, **attrs):
    mod = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(mod, k, v)
    sys.modules[name] = mod
    return mod


def install():
    """Make ``import networks.*`` / ``import utils.*`` resolve to the reference."""
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError(f"reference tree not found at {REF_ROOT}")
    sys.dont_write_bytecode = True
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    import torch
    from torch import nn

    if "kornia" not in sys.modules:
        _stub("kornia", create_meshgrid=lambda *a, **k: (_ for _ in ()).throw(NotImplementedError()))
    for name in ("cv2", "h5py", "imageio"):
        if name not in sys.modules:
            _stub(name)
    if "numba" not in sys.modules:
        def njit(*a, **k):
            if len(a) == 1 and callable(a[0]) and not k:
                return a[0]
            return lambda f: f
        _stub("numba", njit=njit, jit=njit)
    if "configargparse" not in sys.modules:
        class ArgumentParser(argparse.ArgumentParser):
            def add_argument(self, *a, **k):
                k.pop("is_config_file", None)
                k.pop("required", None)
                return super().add_argument(*a, **k)
        _stub("configargparse", ArgumentParser=ArgumentParser)

    # hard-coded .cuda() calls (voxnerf.py:86, tonemapping.py:147, renderer.py:609)
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self

    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)


def blurfactory_args(**overrides):
    """argparse-like namespace with the values of the blurfactory config that reach the
    renderer (configs/evdeblurnerf_blender/tx_blurfactory_evdeblurnerf_ediprior_evcrf.txt)."""
    import torch
    ns = types.SimpleNamespace(
        multires=10, multires_views=4, use_viewdirs=True, mode="c2f",
        kernel_type="RBK", kernel_use_awp=False, N_importance=64, N_samples=64,
        netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256,
        rgb_activate="sigmoid", rgb_add_bias=False, sigma_activate="relu",
        render_rmnearplane=0,
        bounding_box=(torch.tensor([-1.5, -1.5, -1.0]), torch.tensor([1.5, 1.5, 1.0])),
        coarse_num_layers=2, coarse_num_layers_color=3, coarse_hidden_dim=64,
        coarse_hidden_dim_color=64, coarse_app_dim=32, coarse_app_n_comp=[64, 16, 16],
        coarse_n_voxels=16777248, coarse_app_actfn="none", kernel_feat_cnl=15,
        fine_num_layers=2, fine_num_layers_color=3, fine_hidden_dim=256,
        fine_hidden_dim_color=256, fine_geo_feat_dim=128, fine_app_dim=32,
        fine_app_n_comp=[64, 16, 16], fine_n_voxels=134217984, fine_app_actfn="none",
    )
    for k, v in overrides.items():
        setattr(ns, k, v)
    return ns


def load_np_state_dict(module, sd, prefix=""):
    """Load a dict of numpy arrays (evdeblurnerf_amd.weights) into a reference module."""
    import torch
    own = module.state_dict()
    new = {}
    for k in own:
        key = prefix + k
        if key not in sd:
            raise KeyError(f"missing parameter {key}")
        t = torch.from_numpy(sd[key].copy())
        if tuple(t.shape) != tuple(own[k].shape):
            raise ValueError(f"shape mismatch for {key}: {tuple(t.shape)} vs {tuple(own[k].shape)}")
        new[k] = t
    module.load_state_dict(new)
    return module
